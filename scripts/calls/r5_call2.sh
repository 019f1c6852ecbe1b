#!/bin/bash
# round 5, GPU call 2: full GPU suite (new: full-depth parity on the timed models, -sm layer on logical devices, the F32-mask + cast graph), the bench line with
# its parity object, A/B of the logits copy kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/parity_log.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -32 gpurun_out/pytest_gpu.log | cut -c1-250
grep "parity-full-depth\|layer-split" gpurun_out/parity_log.txt | cut -c1-700
echo "== bench"
timeout 600 python bench.py > gpurun_out/r5c2_bench.json 2> gpurun_out/r5c2_bench.err; tail -3 gpurun_out/r5c2_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c2_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'prefill_tok_s', 'hipGraphLaunch_host_us', 'graph_compute_host_us_per_step', 'host_us_per_step', 'parity', 'roofline')})
print(d.get('cpu_baseline'))
PY
echo "== A/B logits copy kernel"
bash scripts/ab_env.sh GGML_MI355X_SMALL_DOWNLOADS 0 1 2>&1 | cut -c1-200
