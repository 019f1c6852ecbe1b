#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernel_classes_us'))"; }
for mc in 33 9; do echo "== GGML_MI355X_Q80_MIN_COLS=$mc llama3-8b-q8_0 -np 32"; GGML_MI355X_Q80_MIN_COLS=$mc one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32; done
