#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "skinny or sibling or np_batch or qkv" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "np32" 2>&1 | tail -5
scripts/ab_env.sh GGML_MI355X_SKINNY_MIX 0 1 -- --preset llama3-8b-q4_k_m --np 32 --prefill 128 2>&1 | tee gpurun_out/ab_mix.txt
python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 64 --preset llama3-8b-q4_k_m --np 32 --prefill 128 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_us_per_step'), d.get('prefill_tok_s'))" | tee -a gpurun_out/ab_mix.txt
