#!/bin/bash
# A/B of one environment knob on the default decode bench line, interleaved, same box:  scripts/ab_env.sh VAR v1 v2 ... [-- extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "${1:-}" == "--" ] && shift
python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 32 "$@" > /dev/null 2>&1  # discarded warm-up
for rep in 1 2; do
  for v in "${VALS[@]}"; do
    env $VAR=$v python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 16 --steps 64 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernel_classes_us', {})
print('$VAR=$v rep $rep: %.1f tok/s %.4f ms/step prefill %.0f | ' % (d['value'], d['ms_per_step'], d.get('prefill_tok_s') or 0) + ' '.join('%s=%.2f' % (a.replace('mmvq_', '').replace('qkv_fused_', 'qkv_'), b) for a, b in sorted(k.items())))"
  done
done
