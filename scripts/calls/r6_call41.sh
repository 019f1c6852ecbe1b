#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tok_s'), d.get('prefill_roofline', {}).get('frac'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'q8_0' in k})"; }
echo "== llama3-8b-q8_0 batch 1 (prefill 2048 in ubatches of 512)"; one --preset llama3-8b-q8_0 --prefill 2048 --steps 32
echo "== llama3-8b-q8_0, ubatch 128"; one --preset llama3-8b-q8_0 --prefill 2048 --steps 32 --ubatch 128
echo "== llama3-8b-q8_0, ubatch 128, skinny up to 128 columns"; GGML_MI355X_Q80_SKINNY_MAX=128 one --preset llama3-8b-q8_0 --prefill 2048 --steps 32 --ubatch 128
echo "== tinyllama prefill 2048"; one --preset tinyllama-1.1b-q8_0 --prefill 2048 --steps 32
echo "== llama3-8b-q4_k_m ubatch 128 (for scale)"; one --prefill 2048 --steps 32 --ubatch 128
