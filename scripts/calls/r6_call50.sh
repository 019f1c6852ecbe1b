#!/bin/bash
# RMS_NORM * w and SwiGLU leaving Q8_0 panel blocks for the 9..128-column kernel (GGML_MI355X_Q80_PRODUCERS): tests, A/B with the parity numbers side by side
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q -m gpu -k "q8_0 or panel or batch_shapes or continuous or verification or rms or glu" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; cb=p.get('continuous_batch') or {}; print(d['value'], d['ms_per_step'], cb.get('within_bar'), cb.get('max_abs'), cb.get('nmse'), sorted((d.get('kernel_classes_us') or {}).keys()))" | cut -c1-500; }
for w in 0 1 0 1; do echo "== GGML_MI355X_Q80_PRODUCERS=$w tinyllama -np 32"; GGML_MI355X_Q80_PRODUCERS=$w one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for w in 0 1; do echo "== GGML_MI355X_Q80_PRODUCERS=$w llama3-8b-q8_0 -np 32"; GGML_MI355X_Q80_PRODUCERS=$w one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32; done
