#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "skinny or sibling or np_batch or qkv" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "np32" 2>&1 | tail -5
scripts/ab_env.sh GGML_MI355X_SKINNY_MIX 1 -- --preset llama3-8b-q4_k_m --np 32 --prefill 128 2>&1 | tee gpurun_out/ab_q6.txt
