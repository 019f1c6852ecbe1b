#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprofv3 kernel stats. Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_log.txt
echo "== rocminfo =="; (rocminfo | grep -E "Marketing Name|gfx9" | head -4) 2>&1
if [ -n "${NODE_DIFF:-}" ]; then
echo "== node diff =="
timeout 300 python scripts/node_diff.py $NODE_DIFF > gpurun_out/node_diff.log 2>&1; grep -E "<<<<" gpurun_out/node_diff.log | head -20; tail -n 3 gpurun_out/node_diff.log
timeout 300 python scripts/node_diff.py $NODE_DIFF 0 > gpurun_out/node_diff_nofuse.log 2>&1; grep -E "<<<<" gpurun_out/node_diff_nofuse.log | head -20
fi
echo "== pytest -m gpu =="
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 45 gpurun_out/pytest_gpu.log | cut -c1-220
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log
echo "== bench =="
timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
echo "== rocprofv3 kernel stats =="
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 32 --warmup 4 --prefill 256 --timing-steps 8 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "rocprof rc=$?"
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 25 "$f"
# keep only the small summaries (the trace itself can be tens of MB)
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
fi
