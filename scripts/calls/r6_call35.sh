#!/bin/bash
# Q8_0 weights: 16 / 32 columns in one pass of the multi-column mat-vec kernel; tests, TinyLlama -np 32 / 16 and a Llama-3-8B Q8_0 -np 32 A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wide_column or mul_mat_q" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'))"; }
for w in 0 1 0 1; do echo "== GGML_MI355X_Q80_WIDE_COLS=$w tinyllama -np 32"; GGML_MI355X_Q80_WIDE_COLS=$w one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_Q80_WIDE_COLS=$w tinyllama -np 16"; GGML_MI355X_Q80_WIDE_COLS=$w one --preset tinyllama-1.1b-q8_0 --np 16 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_Q80_WIDE_COLS=$w llama3-8b-q8_0 -np 32"; GGML_MI355X_Q80_WIDE_COLS=$w one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32 --no-cpu-baseline; done
echo "== llama3-8b-q8_0 batch 1"; one --preset llama3-8b-q8_0 --prefill 2048 --steps 64 --no-cpu-baseline
