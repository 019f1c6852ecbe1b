#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'q8_0' in k})"; }
for pf in 1 0 1 0; do echo "== GGML_MI355X_Q80_SKINNY_PF=$pf llama3-8b-q8_0 -np 32"; GGML_MI355X_Q80_SKINNY_PF=$pf one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32; done
for pf in 1 0; do echo "== GGML_MI355X_Q80_SKINNY_PF=$pf tinyllama -np 32"; GGML_MI355X_Q80_SKINNY_PF=$pf one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
