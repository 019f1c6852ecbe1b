#!/bin/bash
# round 4, GPU call 1: full GPU suite (new TP8-shard / draft-shape / 70B-split / bench dry-run tests) + headline + draft-shape bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export MI355X_PARITY_LOG=gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -15
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json
for cfg in "--np 32 --draft 4" "--np 32 --draft 1" "--np 1 --draft 8" "--np 1 --draft 16" "--np 32"; do
  timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 --prefill 128 --steps 64 $cfg >> gpurun_out/draft_configs.jsonl 2>> gpurun_out/draft_configs.err
done
cut -c1-400 gpurun_out/draft_configs.jsonl
