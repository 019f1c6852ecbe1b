#!/bin/bash
# round 5, GPU call 1: the graph key compared in place instead of the byte-serial FNV fingerprint — tests that exercise graph
# replay, A/B of the round-4 library against this one on ONE box, and the replayed-step timeline with the host's turn on its own
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py -m gpu -x -q 2>&1 | tail -8
echo "== A/B (ab/a_r04.so = round 4's HEAD, ab/b_new.so = key compared in place)"
REPS=2 bash scripts/ab_decode.sh 2>&1 | cut -c1-300
echo "== bench line (new)"
timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 > gpurun_out/r5c1_bench.json 2> gpurun_out/r5c1_bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c1_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'prefill_tok_s', 'hipGraphLaunch_host_us', 'graph_compute_host_us_per_step', 'host_us_per_step')})
PY
echo "== timeline, old then new"
for so in a_r04 b_new; do
  rm -rf gpurun_out/tl_$so
  GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/$so.so timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl_$so -o t --output-format csv -- python bench.py --steps 48 --warmup 4 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0 > gpurun_out/tl_$so.json 2> gpurun_out/tl_$so.err
  f=$(find gpurun_out/tl_$so -name "*kernel_trace.csv" | head -1)
  echo "-- $so: $(grep -o '"value": [0-9.]*' gpurun_out/tl_$so.json | head -1)"
  [ -n "$f" ] && python scripts/decode_gaps.py $f | tail -14 > gpurun_out/r5c1_gaps_$so.txt; cat gpurun_out/r5c1_gaps_$so.txt
  rm -rf gpurun_out/tl_$so
done
