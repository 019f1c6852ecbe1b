#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "np_batch or qkv or rope or skinny" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity_e2e.py -m gpu -x -q -k "np32 or np or batch" 2>&1 | tail -4
scripts/ab_env.sh GGML_MI355X_SKINNY_MIX 1 -- --preset llama3-8b-q4_k_m --np 32 --prefill 128 2>&1 | cut -c1-420 | tee gpurun_out/ab_ropetab.txt
