#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/prof_fa0
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fa0 -o np32fa0 -- python bench.py --preset llama3-8b-q4_k_m --np 32 --fa 0 --prefill 128 --steps 64 --no-cpu-baseline --timing-steps 0 > gpurun_out/np32_fa0_under_rocprof.json 2> gpurun_out/np32_fa0_prof.err; echo "rc=$?"
python scripts/prof_summary.py gpurun_out/prof_fa0/np32fa0_results.db > gpurun_out/np32_fa0_kernel_stats.csv; head -n 16 gpurun_out/np32_fa0_kernel_stats.csv | cut -c1-150
find gpurun_out/prof_fa0 -size +20M -delete
