#!/bin/bash
# round 6, GPU call 4: the decode copy (plane layout + non-temporal loads) and the one-launch step head, built: tests, then A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py tests/test_gpu_baseline_shapes.py tests/test_gpu_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12 | cut -c1-300
grep -h "parity-full-depth\|decode copy" gpurun_out/parity_log.txt | cut -c1-420
echo "== A/B decode copy"
bash scripts/ab_env.sh GGML_MI355X_DECODE_COPY 0 1 2>&1 | cut -c1-420 | tee gpurun_out/r6c4_ab_decode_copy.txt
echo "== A/B step head"
bash scripts/ab_env.sh GGML_MI355X_STEP_HEAD 0 1 2>&1 | cut -c1-200 | tee gpurun_out/r6c4_ab_step_head.txt
echo "== bench"
timeout 600 python bench.py --steps 64 > gpurun_out/r6c4_bench.json 2> gpurun_out/r6c4_bench.err; tail -c 1800 gpurun_out/r6c4_bench.json; tail -3 gpurun_out/r6c4_bench.err
