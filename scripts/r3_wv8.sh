#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "flash or attn" 2>&1 | tail -4
GGML_MI355X_FA_LIST_WV8=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "flash or attn or np32" 2>&1 | tail -4
scripts/ab_env.sh GGML_MI355X_FA_LIST_WV8 0 1 -- --preset llama3-8b-q4_k_m --np 32 --prefill 128 2>&1 | cut -c1-120 | tee gpurun_out/ab_list_wv8.txt
scripts/ab_env.sh GGML_MI355X_FA_WV8 0 1 2>&1 | cut -c1-120 | tee gpurun_out/ab_wv8c.txt
