#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kv_types.py -x -q -m gpu -k "flash_attn or kv" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -3
