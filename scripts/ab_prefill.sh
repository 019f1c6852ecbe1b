#!/bin/bash
# A/B of library builds on ONE box: a prefill-heavy bench line for each llama_box_amd/ab/*.so, REPS rounds interleaved, after one discarded
# warm-up run (the first run of a call pages the image in and reads 2-4 % low)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFG=${CFG:---preset llama3-8b-q4_k_m --prefill 2048}
one() { GGML_BACKEND_PATH=$PWD/$1 timeout -k 5 200 python bench.py $CFG --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', 'prefill', d.get('prefill_tok_s'), 'decode', d['value'])"; }
one $(ls llama_box_amd/ab/*.so | head -1) > /dev/null
for rep in $(seq ${REPS:-2}); do for so in llama_box_amd/ab/*.so; do one $so; done; done
