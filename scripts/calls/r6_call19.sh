#!/bin/bash
# the decode copy for rows that end in a short group (Qwen2-7B: 14 / 74 super-blocks): layout test, bit-equality, Qwen2 lines A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "decode_copy or mat_vecs_over or no_room" 2>&1 | tail -5
true
for dc in 1 0; do
  echo "== GGML_MI355X_DECODE_COPY=$dc qwen2 8064"
  GGML_MI355X_DECODE_COPY=$dc python bench.py --preset qwen2-7b-q5_k_m --prefill 8064 --steps 48 --warmup 4 --pmc-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('parity') or {}).get('within_bar'), d.get('decode_copy'))"
  echo "== GGML_MI355X_DECODE_COPY=$dc qwen2 512"
  GGML_MI355X_DECODE_COPY=$dc python bench.py --preset qwen2-7b-q5_k_m --prefill 512 --steps 48 --warmup 4 --pmc-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('parity') or {}).get('within_bar'), d.get('decode_copy'))"
done
python bench.py --steps 64 --warmup 4 --no-cpu-baseline --pmc-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline']['frac'])"
