#!/bin/bash
# round 5, GPU call 4: -sm layer on 2 / 4 logical devices, the tensor-parallel tests incl. the failure drill, the bench line with the quota-aware CPU leg,
# one replayed decode step launch by launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_tp_p2p.py -m gpu -x -q --durations=8 2>&1 | tail -25 | cut -c1-300
grep "layer-split\|tp-p2p drill" gpurun_out/parity_log.txt | cut -c1-600 | tail -12
echo "== bench"
timeout 600 python bench.py > gpurun_out/r5c4_bench.json 2> gpurun_out/r5c4_bench.err; tail -3 gpurun_out/r5c4_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c4_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'prefill_tok_s', 'hipGraphLaunch_host_us', 'graph_compute_host_us_per_step', 'parity')})
print(d.get('cpu_baseline'))
PY
echo "== one replayed step"
rm -rf gpurun_out/tl; timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o t --output-format csv -- python bench.py --steps 48 --warmup 4 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0 > gpurun_out/tl.json 2> gpurun_out/tl.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1); python scripts/decode_gaps.py $f -v > gpurun_out/r5c4_step.txt; tail -24 gpurun_out/r5c4_step.txt; rm -rf gpurun_out/tl
