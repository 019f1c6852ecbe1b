#!/bin/bash
# long contexts: 32 k tokens (f16 and q8_0 caches), decode + parity object; Qwen2 at 32 k
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 900 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --cpu-steps 8 2>gpurun_out/r6c24.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:90], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tok_s'), (d.get('parity') or {}).get('within_bar'))" || tail -5 gpurun_out/r6c24.err; }
one --prefill 32000 --steps 64
one --prefill 32000 --steps 64 --ctkv q8_0
one --prefill 32000 --steps 64 --fa 0
one --preset qwen2-7b-q5_k_m --prefill 32000 --steps 64
one --np 32 --prefill 1000 --steps 64
