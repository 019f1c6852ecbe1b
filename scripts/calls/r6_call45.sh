#!/bin/bash
# -np 2 / 3: the multi-column mat-vec path (unfused: 17 launches a layer) against the weight-streaming matrix-core path from 2 columns on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'))"; }
for mc in 3 2 3 2; do echo "== GGML_MI355X_MMQ_MIN_COLS=$mc -np 2"; GGML_MI355X_MMQ_MIN_COLS=$mc one --np 2 --prefill 128 --steps 64; done
for mc in 3 2; do echo "== GGML_MI355X_MMQ_MIN_COLS=$mc -np 2 -fa 0"; GGML_MI355X_MMQ_MIN_COLS=$mc one --np 2 --fa 0 --prefill 128 --steps 64; done
for mc in 3 2; do echo "== GGML_MI355X_MMQ_MIN_COLS=$mc qwen2 -np 2"; GGML_MI355X_MMQ_MIN_COLS=$mc one --preset qwen2-7b-q5_k_m --np 2 --prefill 128 --steps 64; done
for mc in 3 2; do echo "== GGML_MI355X_MMQ_MIN_COLS=$mc 1 sequence + 1 draft"; GGML_MI355X_MMQ_MIN_COLS=$mc one --np 1 --draft 1 --prefill 128 --steps 64; done
echo "== -np 3"; one --np 3 --prefill 128 --steps 64
