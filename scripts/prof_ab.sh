#!/bin/bash
# A/B profiling of backend options: usage: prof_ab.sh "ENVA=.. " "ENVB=.."  (each arg is an env assignment string)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); rm -rf gpurun_out/prof_$i
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$i -o bench -- python bench.py --steps 32 --warmup 4 --prefill 256 --timing-steps 0 --no-cpu-baseline > gpurun_out/prof_$i.json 2> gpurun_out/prof_$i.err
  echo "== $cfg"; grep -o '"value": [0-9.]*' gpurun_out/prof_$i.json | head -1
  python scripts/prof_summary.py gpurun_out/prof_$i/bench_results.db | grep -v "k_mmq\|8, 1, false" | head -${TOPN:-10} | cut -c1-120
  find gpurun_out/prof_$i -name "*.db" -size +20M -delete
done
