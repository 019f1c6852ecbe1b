#!/bin/bash
# round 6, GPU call 6: (1) timeline inside the decode attention split kernel (stamps), (2) attention tests after the division-free preamble / split cap,
# (3) A/B of the 4-wave form at 8 k, (4) the whole GPU suite for regressions
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== stamps at 2 k and 8 k"
GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/fa_stamp.so N_WG=192 timeout 300 python scripts/lab/fa_stamps.py 2048 2>&1 | tail -9 | tee gpurun_out/r6c6_fa_stamps.txt
GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/fa_stamp.so N_WG=256 timeout 300 python scripts/lab/fa_stamps.py 7936 2>&1 | tail -9 | tee -a gpurun_out/r6c6_fa_stamps.txt
echo "== 8-wave vs 4-wave form by context (after the split cap)"
one() { env GGML_MI355X_FA_WV8=$2 GGML_MI355X_FA_SPLITS=$3 python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 --steps 64 --prefill $1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d.get('kernel_classes_us', {})
print('prefill $1 wv8 $2 splits $3: %.1f tok/s %.4f ms/step flash_attn=%.2f' % (d['value'], d['ms_per_step'], k.get('flash_attn', 0)))"; }
one 2048 1 0 > /dev/null
for rep in 1 2; do one 2048 1 0; one 2048 0 0; one 2048 0 36; one 7936 1 0; one 7936 0 0; one 7936 0 64; one 7936 0 96; done | tee gpurun_out/r6c6_fa_forms.txt
echo "== full GPU suite"
rm -f gpurun_out/parity_log.txt
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 2>&1 | tail -16 | cut -c1-250
