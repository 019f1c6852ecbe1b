#!/bin/bash
# round 6, GPU call 15: VERDICT r05 #5 (b) — the combine as the tail of the split launch (merge2) at FEW fat splits on 8-wave groups, 2 k and 8 k context
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { env GGML_MI355X_FA_SELF_MERGE=$2 GGML_MI355X_FA_SPLITS=$3 python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 --steps 64 --prefill $1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernel_classes_us', {})
print('prefill $1 self_merge $2 splits $3: %.1f tok/s %.4f ms/step flash_attn=%.2f' % (d['value'], d['ms_per_step'], k.get('flash_attn', 0)))"; }
one 2048 0 0 > /dev/null
for rep in 1 2; do one 2048 0 0; for sp in 2 3 4 8 16 24; do one 2048 1 $sp; done; done | tee gpurun_out/r6c15_merge2_few_splits.txt
one 7936 0 0 | tee -a gpurun_out/r6c15_merge2_few_splits.txt; for sp in 4 8 16 32; do one 7936 1 $sp; done | tee -a gpurun_out/r6c15_merge2_few_splits.txt
