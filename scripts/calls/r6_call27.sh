#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash_attn" 2>&1 | tail -3
grep "D=64" gpurun_out/parity_log.txt | cut -c1-220 | tail -24
# Llama-3.2-1B-like shape (head_dim 64, 32 heads / 8 KV heads) on the model path: the test preset at these head counts, decode vs the oracle
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "greedy or decode" 2>&1 | tail -3
