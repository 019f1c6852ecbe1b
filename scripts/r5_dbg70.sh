cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { env "$@" GPU_MAX_HW_QUEUES=16 timeout 300 python tests/split_worker.py model 70b 2>&1 | grep "^SPLIT_JSON\|mi355x:" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('SPLIT_JSON'):
        c = json.loads(l[11:])['cases'][0]; print('  $*'[:120], 'row0 %.0e row1 %.0e' % (c['nmse_rows_vs_one_device'][0], c['nmse_rows_vs_one_device'][1]))
    else: print('  ', l.strip()[:300])
"; }
for i in 1 2 3; do
run GGML_MI355X_FAKE_DEVICES=8 GGML_MI355X_SPLIT_GRAPHS=0 GGML_MI355X_SPLIT_THREADS=1 GGML_MI355X_DBG_SUBMIT_ORDER=desc
run GGML_MI355X_FAKE_DEVICES=8 GGML_MI355X_SPLIT_GRAPHS=1 GGML_MI355X_SPLIT_THREADS=1
done
