#!/bin/bash
# Q8_0 weights, 9..32 columns: 8-column mat-vec passes (the weights streamed once per pass) against the matrix-core GEMM of the prompt batches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 400 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for mc in 33 9 33 9; do echo "== GGML_MI355X_Q80_MIN_COLS=$mc tinyllama -np 32"; GGML_MI355X_Q80_MIN_COLS=$mc one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for mc in 33 9; do echo "== GGML_MI355X_Q80_MIN_COLS=$mc tinyllama -np 16"; GGML_MI355X_Q80_MIN_COLS=$mc one --preset tinyllama-1.1b-q8_0 --np 16 --prefill 128 --steps 64; done
for mc in 33 9; do echo "== GGML_MI355X_Q80_MIN_COLS=$mc tinyllama -np 12"; GGML_MI355X_Q80_MIN_COLS=$mc one --preset tinyllama-1.1b-q8_0 --np 12 --prefill 128 --steps 64; done
