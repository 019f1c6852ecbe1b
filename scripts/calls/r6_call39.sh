#!/bin/bash
# Q8_0 9..32-column kernel over panel layouts (weights copy + activations): tests, A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "9_to_32 or mul_mat_q" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "q8_0 or decode_copy or panel" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny or continuous or batch_shapes" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'q8_0' in k}, d.get('decode_copy'))"; }
for w in 0 1 0 1; do echo "== GGML_MI355X_DECODE_COPY=$w tinyllama -np 32"; GGML_MI355X_DECODE_COPY=$w one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_DECODE_COPY=$w llama3-8b-q8_0 -np 32"; GGML_MI355X_DECODE_COPY=$w one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32 --no-cpu-baseline; done
echo "== tinyllama -np 16"; one --preset tinyllama-1.1b-q8_0 --np 16 --prefill 128 --steps 64
