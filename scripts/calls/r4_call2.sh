#!/bin/bash
# round 4, GPU call 2: baseline-shapes tests (list attention at draft shapes, attn_nf gate), draft bench lines before/after
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_ops.py -x -q -k "batch or non_flash or flash_attn" 2>&1 | tail -15
for cfg in "--np 32 --draft 4" "--np 32 --draft 1"; do
  for v in 0 1; do
    GGML_MI355X_FA_SPARSE_LISTS=$v timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 --prefill 128 --steps 64 $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_classes_us']
print('sparse_lists=$v $cfg: %.3f ms/step value %.0f | flash_attn %.1f us | host %s' % (d['ms_per_step'], d['value'], k.get('flash_attn', 0), d['host_us_per_step']))" | tee -a gpurun_out/draft_ab.txt
  done
done
