#!/bin/bash
# A/B of library builds on ONE box, decode: the default bench line for each llama_box_amd/ab/*.so, REPS rounds interleaved, after a warm-up
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() { GGML_BACKEND_PATH=$PWD/$1 timeout -k 5 200 python bench.py ${CFG:-} --steps 128 --no-cpu-baseline --pmc-traffic 0 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_classes_us'].items() if 'mmvq' in k})"; }
one $(ls llama_box_amd/ab/*.so | head -1) > /dev/null
for rep in $(seq ${REPS:-2}); do for so in llama_box_amd/ab/*.so; do one $so; done; done
