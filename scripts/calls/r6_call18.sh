#!/bin/bash
# round 6, GPU call 18: kernel stats of the secondary configurations (looking for launches that should not be there)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
prof() { tag=$1; shift; rm -rf gpurun_out/pp; timeout 500 rocprofv3 --kernel-trace -d gpurun_out/pp -o b --output-format csv -- python bench.py "$@" --steps 48 --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 > /dev/null 2>&1
  f=$(find gpurun_out/pp -name "*kernel_trace.csv" | head -1); echo "== $tag: $*"; python scripts/trace_stats.py $f | head -${TOP:-16} | cut -c1-150; rm -rf gpurun_out/pp; }
prof fa0_np32 --fa 0 --np 32 --prefill 128 | tee gpurun_out/r6c18_stats.txt
prof qwen_8k --preset qwen2-7b-q5_k_m --prefill 8064 | tee -a gpurun_out/r6c18_stats.txt
prof tinyllama --preset tinyllama-1.1b-q8_0 --prefill 512 | tee -a gpurun_out/r6c18_stats.txt
prof q8_0_kv --ctkv q8_0 --prefill 2048 | tee -a gpurun_out/r6c18_stats.txt
