#!/bin/bash
# Q8_0 weights at 9..32 columns on the matrix cores (k_mmq_q80_skinny): tests, TinyLlama -np 32/16 and Llama-3-8B Q8_0 -np 32 A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "9_to_32 or mul_mat_q" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny or continuous or batch_shapes" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'q8_0' in k})"; }
for w in 0 1 0 1; do echo "== GGML_MI355X_Q80_SKINNY=$w tinyllama -np 32"; GGML_MI355X_Q80_SKINNY=$w one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_Q80_SKINNY=$w tinyllama -np 16"; GGML_MI355X_Q80_SKINNY=$w one --preset tinyllama-1.1b-q8_0 --np 16 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_Q80_SKINNY=$w llama3-8b-q8_0 -np 32"; GGML_MI355X_Q80_SKINNY=$w one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32 --no-cpu-baseline; done
