#!/bin/bash
# round 6, GPU call 9: a bf16 cache read in place; the Q5_K fused Q/K/V forms without scratch (Qwen2); kv tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kv_types.py tests/test_gpu_ops.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250
one() { python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 --steps 64 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernel_classes_us', {})
print('$*: %.1f tok/s %.4f ms/step' % (d['value'], d['ms_per_step']), {a: b for a, b in k.items() if 'flash' in a or 'qkv' in a or 'image' in a})"; }
for rep in 1 2; do
one --prefill 2048 --ctkv bf16; one --prefill 7936 --ctkv bf16
one --preset qwen2-7b-q5_k_m --prefill 8064
done | tee gpurun_out/r6c9_bf16_qwen.txt
