#!/bin/bash
# any 2..8 query heads per KV head on the lane-parallel attention kernel (+ the generic kernel's new instantiations): attention tests of all three files, model tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kv_types.py -x -q -m gpu -k "flash_attn or kv" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py -x -q -m gpu 2>&1 | tail -3
grep "H=24/8\|H=12/2\|H=20/4\|H=14/2" gpurun_out/parity_log.txt | cut -c1-200 | tail -40
