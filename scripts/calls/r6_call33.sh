#!/bin/bash
# other families' shapes: Llama-2-7B (multi-head attention), Llama-3.2-3B (3 query heads per KV head), Llama-3.2-1B (head_dim 64, 4 per KV head); bench lines with parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --cpu-steps 8 2>gpurun_out/r6c33.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['config']['workload'][:60], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tok_s'), (p.get('continuous_batch') or p).get('within_bar'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'attn' in k}, d.get('decode_copy', {}).get('tensors'))" || tail -3 gpurun_out/r6c33.err; }
for gm in 2 1; do echo "== GGML_MI355X_FA_G_MIN=$gm llama2-7b"; GGML_MI355X_FA_G_MIN=$gm one --preset llama2-7b-q4_k_m --prefill 2048 --steps 64; done
echo "== llama3.2-3b"; one --preset llama3.2-3b-q4_k_m --prefill 2048 --steps 64
echo "== llama3.2-3b -np 32"; one --preset llama3.2-3b-q4_k_m --np 32 --prefill 128 --steps 64
echo "== llama3.2-1b"; one --preset llama3.2-1b-q4_k_m --prefill 2048 --steps 64
echo "== llama3.2-1b -np 32"; one --preset llama3.2-1b-q4_k_m --np 32 --prefill 128 --steps 64
echo "== llama2-7b -fa 0"; one --preset llama2-7b-q4_k_m --fa 0 --prefill 2048 --steps 64
