#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "qkv or rope or fused" 2>&1 | grep -E "^E|Error|assert|FAILED" | head -20
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "decode or layer" 2>&1 | grep -E "^E|Error|assert|FAILED" | head -20
