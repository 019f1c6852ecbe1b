#!/bin/bash
# what a re-capture is made of (host time): walk into the capture, executable-graph update, and the eager walk of a shadow capture
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 400 python bench.py "$@" --pmc-traffic 0 --no-cpu-baseline --timing-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('graph_compute_host_us_per_step'))"; }
for sh in 0 1; do
  echo "== GGML_MI355X_SHADOW_CAPTURE=$sh: -np 32"; GGML_MI355X_SHADOW_CAPTURE=$sh one --np 32 --prefill 128 --steps 64
  echo "== GGML_MI355X_SHADOW_CAPTURE=$sh: batch 1 across a boundary"; GGML_MI355X_SHADOW_CAPTURE=$sh one --prefill 2000 --steps 128
done
