#!/bin/bash
# inputs of DESIGN.md §6's time budget: per-rank compute of Llama-3-70B under an 8-way (and 4-, 2-way) tensor split and of Llama-3-8B under tp8 / tp2,
# each timed as ONE rank's shard alone on one GPU (no collectives), beside the unsharded model; and the p2p test (all-reduce kernel on a shared GPU)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; : > gpurun_out/tp_budget.jsonl
run() { echo "== $*"; timeout 600 python bench.py "$@" --no-cpu-baseline --timing-steps 8 --pmc-traffic 0 2>> gpurun_out/tp_budget.err | tee -a gpurun_out/tp_budget.jsonl | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   %.4f ms/step prefill %s tok/s | %s' % (d['ms_per_step'], d.get('prefill_tok_s'), ' '.join('%s=%.2f' % (k.replace('mmvq_', '').replace('qkv_fused_', 'qkv_'), v) for k, v in sorted(d['kernel_classes_us'].items()))))"; }
run --preset llama3-70b-q4_k_m --prefill 512 --steps 32
run --preset llama3-70b-q4_k_m --prefill 512 --steps 32 --emulate-tp 8
run --preset llama3-70b-q4_k_m --prefill 512 --steps 32 --emulate-tp 4
run --preset llama3-70b-q4_k_m --prefill 512 --steps 32 --emulate-tp 2
run --preset llama3-8b-q4_k_m --prefill 2048 --steps 64 --emulate-tp 8
run --preset llama3-8b-q4_k_m --prefill 2048 --steps 64 --emulate-tp 2
