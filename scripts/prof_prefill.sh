#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_pf
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pf -o bench -- python bench.py --steps 4 --warmup 2 --prefill 2048 --timing-steps 0 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/prof_pf.json 2> gpurun_out/prof_pf.err
grep -o '"prefill_tok_s": [0-9.]*' gpurun_out/prof_pf.json
python scripts/prof_summary.py gpurun_out/prof_pf/bench_results.db | head -${TOPN:-14} | cut -c1-130
find gpurun_out/prof_pf -name "*.db" -size +20M -delete
