#!/bin/bash
# round 6, GPU call 3: labs — (1) line-aligned planes + non-temporal weight loads (same bytes from the addresses a repack would have; wrong results,
# timing only), (2) one decode token's attention combine leaving Q8_K blocks for wo (no prologue in wo), (3) Qwen2 with a q8_0 cache keeps its fused Q/K/V launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== planes / nt lab (ab/*.so; tok/s, ms/step, mat-vec classes)"
REPS=2 bash scripts/ab_decode.sh 2>&1 | cut -c1-420 | tee gpurun_out/r6c3_planes_nt_lab.txt
echo "== combine -> Q8_K -> wo at batch 1"
bash scripts/ab_env.sh GGML_MI355X_FA_Q8_B1 0 1 2>&1 | cut -c1-420 | tee gpurun_out/r6c3_fa_q8_b1.txt
echo "== tests with the knob on"
GGML_MI355X_FA_Q8_B1=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
echo "== qwen2 q8_0 cache at 8 k"
for kv in f16 q8_0; do timeout 300 python bench.py --preset qwen2-7b-q5_k_m --steps 64 --warmup 8 --prefill 7936 --ctkv $kv --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$kv', d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_classes_us'].items() if 'qkv' in k or 'set_rows' in k or 'flash' in k or 'rope' in k})"; done
timeout 300 python -m pytest tests/test_gpu_kv_types.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
