#!/bin/bash
# round 6, GPU call 12: the fused Q/K/V launch of two formats split by WAVES (one unit per wave over 256 workgroups) — tests, then A/B by environment
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_baseline_shapes.py tests/test_gpu_kv_types.py tests/test_gpu_parity_e2e.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
bash scripts/ab_env.sh GGML_MI355X_QKV_WAVE_SPLIT 0 1 2>&1 | cut -c1-330 | tee gpurun_out/r6c12_ab_qkv_wave_split.txt
bash scripts/ab_env.sh GGML_MI355X_QKV_WAVE_SPLIT 0 1 -- --preset qwen2-7b-q5_k_m --prefill 8064 2>&1 | cut -c1-330 | tee -a gpurun_out/r6c12_ab_qkv_wave_split.txt
bash scripts/ab_env.sh GGML_MI355X_QKV_WAVE_SPLIT 0 1 -- --preset llama3-70b-q4_k_m --prefill 512 --steps 32 2>&1 | cut -c1-330 | tee -a gpurun_out/r6c12_ab_qkv_wave_split.txt
