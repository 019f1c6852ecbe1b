#!/bin/bash
# the Q8_0 prompt GEMM over panel-ordered operands: tests, prefill A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "q8_0 or panel or mul_mat_q" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tok_s'), (d.get('parity') or {}).get('within_bar'))"; }
for g in 0 1 0 1; do echo "== GGML_MI355X_Q80_GEMM_PANELS=$g llama3-8b-q8_0"; GGML_MI355X_Q80_GEMM_PANELS=$g one --preset llama3-8b-q8_0 --prefill 2048 --steps 16 --no-cpu-baseline; done
for g in 0 1; do echo "== GGML_MI355X_Q80_GEMM_PANELS=$g tinyllama"; GGML_MI355X_Q80_GEMM_PANELS=$g one --preset tinyllama-1.1b-q8_0 --prefill 2048 --steps 16; done
