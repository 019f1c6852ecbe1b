cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { env "$@" GPU_MAX_HW_QUEUES=16 timeout 600 python tests/split_worker.py model 70b 2>&1 | grep "^SPLIT_JSON\|mi355x:" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('SPLIT_JSON'):
        c = json.loads(l[11:])['cases'][0]; print('  $*'[:170], 'row0 %.0e row1 %.0e' % (c['nmse_rows_vs_one_device'][0], c['nmse_rows_vs_one_device'][1]), c.get('timed_split'), c.get('timed_one_device'))
    else: print('  ', l.strip()[:300])
"; }
for i in 1 2; do
run GGML_MI355X_FAKE_DEVICES=8 SPLIT_TIME=60 SPLIT_LAYERS=16 GGML_MI355X_SPLIT_THREADS=0
run GGML_MI355X_FAKE_DEVICES=8 SPLIT_TIME=60 SPLIT_LAYERS=16
run GGML_MI355X_FAKE_DEVICES=2 SPLIT_TIME=60 SPLIT_LAYERS=16 GGML_MI355X_SPLIT_THREADS=0
run GGML_MI355X_FAKE_DEVICES=2 SPLIT_TIME=60 SPLIT_LAYERS=16
done
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_tp_p2p.py -m gpu -x -q 2>&1 | tail -5
