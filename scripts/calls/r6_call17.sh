#!/bin/bash
# round 6, GPU call 17: the non-flash path (llama-box's default) without its per-layer cont(permute(kqv)) copy at batch 1 — tests, then A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py tests/test_gpu_baseline_shapes.py tests/test_gpu_kv_types.py "tests/test_gpu_full_depth.py::test_full_depth_bar_as_written_on_damped_weights" -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
bash scripts/ab_env.sh GGML_MI355X_ELIDE_CONT 0 1 -- --fa 0 2>&1 | cut -c1-200 | tee gpurun_out/r6c17_ab_elide_cont_fa0.txt
bash scripts/ab_env.sh GGML_MI355X_ELIDE_CONT 0 1 -- --fa 0 --preset qwen2-7b-q5_k_m --prefill 2048 2>&1 | cut -c1-200 | tee -a gpurun_out/r6c17_ab_elide_cont_fa0.txt
