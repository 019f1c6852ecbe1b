#!/bin/bash
# a sweep over configurations nobody has looked at this round: where is something anomalous?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s' % d['config']['workload'][:70], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tok_s'), {k: v for k, v in (d.get('kernel_classes_us') or {}).items()})" | cut -c1-700; }
one --preset qwen2-7b-q5_k_m --np 32 --prefill 128 --steps 32
one --preset qwen2-7b-q5_k_m --fa 0 --prefill 2048 --steps 32
one --np 2 --prefill 128 --steps 32
one --np 4 --prefill 128 --steps 32
one --np 16 --prefill 128 --steps 32
one --preset llama3-70b-q4_k_m --np 32 --prefill 64 --steps 16
one --prefill 2048 --steps 16 --ubatch 128
one --prefill 2048 --steps 16 --ubatch 256
one --prefill 2048 --steps 16 --ubatch 2048 --n-batch 2048
one --np 32 --prefill 128 --steps 32 --ctkv q4_0
one --preset llama3.2-3b-q4_k_m --fa 0 --prefill 2048 --steps 32
