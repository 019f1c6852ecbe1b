#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_e2e.py -x -q -m gpu 2>&1 | tail -2
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --timing-steps 0 --cpu-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity') or {}; print(d['value'], d['ms_per_step'], (p.get('continuous_batch') or p).get('within_bar'))"; }
echo "== -np 2"; one --np 2 --prefill 128 --steps 64
echo "== qwen2 -np 2"; one --preset qwen2-7b-q5_k_m --np 2 --prefill 128 --steps 64
echo "== 70b -np 2"; one --preset llama3-70b-q4_k_m --np 2 --prefill 64 --steps 16 --no-cpu-baseline
echo "== llama3.2-3b -np 2"; one --preset llama3.2-3b-q4_k_m --np 2 --prefill 128 --steps 64
