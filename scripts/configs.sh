#!/bin/bash
# The other BASELINE.json configs (parity cases / secondary numbers), one JSON line each into gpurun_out/configs.jsonl
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; : > gpurun_out/configs.jsonl
run() { echo "== $*"; timeout 600 python bench.py "$@" --no-cpu-baseline --timing-steps 8 --pmc-traffic ${CONFIGS_PMC:-0} 2>> gpurun_out/configs.err | tee -a gpurun_out/configs.jsonl | cut -c1-420; }
run --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 64
run --preset qwen2-7b-q5_k_m --prefill 8064 --steps 64
run --preset tinyllama-1.1b-q8_0 --prefill 512 --steps 128
run --preset llama3-8b-q4_k_m --fa 0 --prefill 2048 --steps 64
run --preset llama3-8b-q4_k_m --fa 0 --np 32 --prefill 128 --steps 64
run --preset llama3-8b-q4_k_m --prefill 2048 --steps 64
run --preset llama3-70b-q4_k_m --prefill 512 --steps 32
run --preset llama3-8b-q4_k_m --np 32 --prefill 512 --steps 64
run --preset llama3-8b-q4_k_m --np 8 --prefill 128 --steps 64
