#!/bin/bash
# whole-group and short-group plane kernels split (T_Q*KP / T_Q*KS): tests, headline, Qwen2 lines, kernel stats of the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "decode_copy or mat_vecs_over or no_room" 2>&1 | tail -3
one() { timeout 400 python bench.py "$@" --pmc-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], (d.get('parity') or {}).get('within_bar'), r.get('kernel'), r.get('avg_us'), r.get('frac'))"; }
echo "== headline x2"; one --steps 128 --warmup 8 --no-cpu-baseline; one --steps 128 --warmup 8 --no-cpu-baseline
echo "== qwen2 8064 / 512"; one --preset qwen2-7b-q5_k_m --prefill 8064 --steps 48 --warmup 4; one --preset qwen2-7b-q5_k_m --prefill 512 --steps 48 --warmup 4 --no-cpu-baseline
rm -rf gpurun_out/pp; timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o b --output-format csv -- python bench.py --steps 128 --warmup 8 --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 > /dev/null 2>&1
f=$(find gpurun_out/pp -name "*kernel_stats.csv" | head -1); head -9 $f | cut -c1-200 | tee gpurun_out/r6c21_headline_stats.txt
rm -rf gpurun_out/pp
