#!/bin/bash
# head_dim 64: the position-list form for 2..32 tokens; op tests, TinyLlama -np 32 A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "flash_attn" 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "tinyllama or TinyLlama or tiny" 2>&1 | tail -3
one() { timeout 400 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'attn' in k})"; }
for on in 0 1 0 1; do echo "== GGML_MI355X_FA_DEC64=$on tinyllama -np 32"; GGML_MI355X_FA_DEC64=$on one --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64; done
for on in 0 1; do echo "== GGML_MI355X_FA_DEC64=$on tinyllama -np 8"; GGML_MI355X_FA_DEC64=$on one --preset tinyllama-1.1b-q8_0 --np 8 --prefill 128 --steps 64; done
