#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "9_to_32 or mul_mat_q" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "q8_0 or panel or batch_shapes or continuous" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'q8_0' in k})"; }
for mx in 32 64; do echo "== GGML_MI355X_Q80_SKINNY_MAX=$mx llama3-8b-q8_0 -np 48"; GGML_MI355X_Q80_SKINNY_MAX=$mx one --preset llama3-8b-q8_0 --np 48 --prefill 128 --steps 32; done
for mx in 32 64; do echo "== GGML_MI355X_Q80_SKINNY_MAX=$mx llama3-8b-q8_0 -np 64"; GGML_MI355X_Q80_SKINNY_MAX=$mx one --preset llama3-8b-q8_0 --np 64 --prefill 128 --steps 32; done
for mx in 32 64; do echo "== GGML_MI355X_Q80_SKINNY_MAX=$mx tinyllama -np 64"; GGML_MI355X_Q80_SKINNY_MAX=$mx one --preset tinyllama-1.1b-q8_0 --np 64 --prefill 128 --steps 32; done
