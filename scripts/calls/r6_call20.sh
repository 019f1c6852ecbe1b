#!/bin/bash
# headline after the short-group generalisation of the plane loads: per-kernel durations against profiles/r06_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do
python bench.py --steps 128 --warmup 8 --no-cpu-baseline --pmc-traffic 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['roofline'])"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 128 --warmup 8 --no-cpu-baseline --pmc-traffic 0 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220
