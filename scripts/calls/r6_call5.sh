#!/bin/bash
# round 6, GPU call 5: decode attention — split count against context (one 8-wave workgroup fills a CU's registers: at 8 k the automatic 64 splits x 8 KV heads
# run as two rounds of one-trip workgroups); A/B through GGML_MI355X_FA_SPLITS
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() { env GGML_MI355X_FA_SPLITS=$2 python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 --steps 64 --prefill $1 --ctkv $3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernel_classes_us', {})
print('prefill $1 kv $3 splits $2: %.1f tok/s %.4f ms/step flash_attn=%.2f' % (d['value'], d['ms_per_step'], k.get('flash_attn', 0)))"; }
one 2048 0 f16 > /dev/null
for rep in 1 2; do
  for sp in 0 16 24 32; do one 2048 $sp f16; done
  for sp in 0 16 32 48; do one 7936 $sp f16; done
done | tee gpurun_out/r6c5_fa_splits.txt
for sp in 0 16 32; do one 7936 $sp q8_0; one 7936 $sp q4_0; done | tee -a gpurun_out/r6c5_fa_splits.txt
echo "== bench line (symbol fix)"
timeout 600 python bench.py --steps 64 > gpurun_out/r6c5_bench.json 2> gpurun_out/r6c5_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6c5_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline'])
PY
