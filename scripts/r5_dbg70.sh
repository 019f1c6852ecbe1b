cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $*"; env "$@" GPU_MAX_HW_QUEUES=16 timeout 300 python tests/split_worker.py model 70b 2>&1 | grep "^SPLIT_JSON\|mi355x:" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('SPLIT_JSON'):
        c = json.loads(l[11:])['cases'][0]; print('  rows', ['%.0e' % x for x in c['nmse_rows_vs_one_device']], 'cache_eq', c['ip']['cache_equal_to_one_device'], 'replays', c['graph_replays'], 'after host write', c['ip']['nmse_after_host_write_vs_one_device'])
    else: print('  ', l.strip()[:300])
"; }
timeout 900 python -m pytest tests/test_gpu_tp_p2p.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
run GGML_MI355X_FAKE_DEVICES=2 GGML_MI355X_SPLIT_GRAPHS=0
run GGML_MI355X_FAKE_DEVICES=8 GGML_MI355X_SPLIT_GRAPHS=0
run GGML_MI355X_FAKE_DEVICES=8 GGML_MI355X_SPLIT_GRAPHS=1
