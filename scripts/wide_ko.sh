#!/bin/bash
# knock-out matrix of the wide prompt GEMM (timing only): prefill tok/s by kernel form (KH) and knock-out (KO).  The switches exist only in the
# lab build: bash scripts/ubench/build_wide_variants.sh, then GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/wide_variants.so bash scripts/wide_ko.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 4 --warmup 1 > /dev/null 2>&1
for kh in ${KHS:-0 2 4}; do for ko in ${KOS:-0 1 2}; do
  r=$(GGML_MI355X_MMQ_WIDE_KH=$kh GGML_MI355X_MMQ_WIDE_KO=$ko python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 --steps 4 --warmup 1 ${BENCH_ARGS:-} 2>/dev/null | grep -o '"prefill_tok_s": [0-9.]*')
  echo "KH=$kh KO=$ko $r"
done; done
