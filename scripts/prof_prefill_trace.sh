#!/bin/bash
# kernel trace of one bench prefill: idle time by follower kernel (prefill_gaps.py) and the per-layer kernel sequence (prefill_shapes.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-pft}
rm -rf gpurun_out/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -o bench -- python bench.py --steps 4 --warmup 2 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0 ${BENCH_ARGS:-} > gpurun_out/prof_$TAG.json 2> gpurun_out/prof_$TAG.err
grep -o '"prefill_tok_s": [0-9.]*' gpurun_out/prof_$TAG.json
F=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
[ "${GAPS:-1}" = "1" ] && python scripts/prefill_gaps.py $F | head -${GAPN:-8} | tee gpurun_out/prefill_gaps_$TAG.txt
python scripts/prefill_shapes.py $F | tee gpurun_out/prefill_shapes_$TAG.txt
find gpurun_out/prof_$TAG -size +1M -delete
