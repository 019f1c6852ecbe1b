#!/bin/bash
# round 6, GPU call 7: the decode attention kernel without batch arithmetic / dependent kernel-argument rounds — tests, stamps again, A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kv_types.py tests/test_gpu_baseline_shapes.py tests/test_gpu_model.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
echo "== stamps"
GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/fa_stamp.so N_WG=192 timeout 300 python scripts/lab/fa_stamps.py 2048 2>&1 | tail -8 | cut -c1-140 | tee gpurun_out/r6c7_fa_stamps.txt
echo "== A/B"
mkdir -p /tmp/abx; mv llama_box_amd/ab/fa_stamp.so /tmp/abx/
CFG="" REPS=2 bash scripts/ab_decode.sh 2>&1 | cut -c1-300 | tee gpurun_out/r6c7_ab_attention_preamble.txt
CFG="--prefill 7936" REPS=1 bash scripts/ab_decode.sh 2>&1 | cut -c1-300 | tee -a gpurun_out/r6c7_ab_attention_preamble.txt
CFG="--np 32 --prefill 128" REPS=1 bash scripts/ab_decode.sh 2>&1 | cut -c1-300 | tee -a gpurun_out/r6c7_ab_attention_preamble.txt
