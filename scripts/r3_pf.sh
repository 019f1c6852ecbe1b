#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { env "$@" python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 16 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*: prefill %.0f tok/s, decode %.1f' % (d['prefill_tok_s'], d['value']))"; }
python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 8 > /dev/null 2>&1
for rep in 1 2; do
  run GGML_MI355X_MMQ_BN=0
  run GGML_MI355X_MMQ_BN=64
  run GGML_MI355X_MMQ_BN=128
  run GGML_MI355X_FA_MMA_MIN_Q=33
done 2>&1 | tee gpurun_out/ab_pf.txt
