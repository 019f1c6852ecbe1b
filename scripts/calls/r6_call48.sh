#!/bin/bash
# Q8_0 sibling mat-muls in one launch (+ the rope / cache-store fusion that follows): tests, TinyLlama / Llama-3-8B Q8_0 -np 32 A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "q8_0 or sibling or mul_mat_q" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "q8_0 or panel or batch_shapes or continuous or verification" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny" 2>&1 | tail -3
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'), d.get('kernel_classes_us'))" | cut -c1-600; }
echo "== tinyllama -np 32"; one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64
echo "== llama3-8b-q8_0 -np 32"; one --preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32 --no-cpu-baseline
