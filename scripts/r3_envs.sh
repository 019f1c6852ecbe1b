#!/bin/bash
# the documented switches of round 3 in their non-default positions: the op tests and the -np 32 layer tests must still pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for e in GGML_MI355X_SKINNY_TP=1 GGML_MI355X_FA_WV8=0 GGML_MI355X_FA_LIST_WV8=1; do
  echo "== $e"; env $e timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "not epilogue and not glu" 2>&1 | tail -3
done 2>&1 | tee gpurun_out/envs.txt
