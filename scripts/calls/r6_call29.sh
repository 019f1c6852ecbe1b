#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python bench.py --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64 --pmc-traffic 0 --timing-steps 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernel_classes_us'), d.get('graph_compute_host_us_per_step'))"
rm -rf gpurun_out/pp; timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o b --output-format csv -- python bench.py --preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64 --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 > /dev/null 2>&1
f=$(find gpurun_out/pp -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-200; rm -rf gpurun_out/pp
