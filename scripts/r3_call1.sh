#!/bin/bash
# round 3, GPU call 1: new parity tests + default bench + loader throughput
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 25 $O/pytest_gpu.log
cp gpurun_out/parity_log.txt $O/parity_log.txt 2>/dev/null
grep -h "baseline-shapes\] .*\(np32\|pf512\|prefill 2048\)" $O/parity_log.txt | cut -c1-260 | tail -n 60
grep -h "config 1\|loader" $O/parity_log.txt | cut -c1-400
for st in 0 1; do GGML_MI355X_STAGED_UPLOAD=$st timeout 300 python scripts/upload_bench.py 2>&1 | tail -n 1; done | tee $O/upload_bench.jsonl
GGML_MI355X_UPLOAD_THREADS=8 timeout 300 python scripts/upload_bench.py 2>&1 | tail -n 1 | tee -a $O/upload_bench.jsonl
GGML_MI355X_UPLOAD_THREADS=1 timeout 300 python scripts/upload_bench.py 2>&1 | tail -n 1 | tee -a $O/upload_bench.jsonl
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | cut -c1-3000
timeout 600 python bench.py --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 64 --no-cpu-baseline --timing-steps 8 > $O/np32.json 2> $O/np32.err; echo "np32 rc=$?"; cat $O/np32.json | cut -c1-2500
