#!/bin/bash
# One gpurun call that produces everything profiles/ holds for a round: parity log, default bench line (roofline with
# PMC traffic + cpu_baseline), rocprofv3 --kernel-trace --stats of the same bench command, a prefill-heavy profile,
# and PMC passes for the dominant kernels.  Outputs land in gpurun_out/collect/ (copied into profiles/ afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/collect; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_log.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 2 $O/pytest_gpu.log
cp gpurun_out/parity_log.txt $O/parity_log.txt 2>/dev/null
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1600 $O/bench.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --no-cpu-baseline --pmc-traffic 0 > $O/bench_under_rocprof.json 2> $O/prof.err; echo "rocprof rc=$?"
python scripts/prof_summary.py gpurun_out/prof/bench_results.db > $O/bench_kernel_stats.csv; head -n 16 $O/bench_kernel_stats.csv | cut -c1-150
find gpurun_out/prof -size +20M -delete
PMC_GROUPS="FETCH_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  BENCH_ARGS="--steps 16 --warmup 2 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0" TOPN=40 bash scripts/prof_pmc.sh > $O/pmc_passes.txt 2>&1
for i in 0 1 2; do cp gpurun_out/pmc_$i.summary.txt $O/pmc_pass$i.csv 2>/dev/null; done
head -n 6 $O/pmc_pass0.csv | cut -c1-160
# the -np 32 configuration: kernel stats of its decode steps (the skinny matrix-core kernel, position-list attention)
rm -rf gpurun_out/prof_np
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_np -o np32 -- python bench.py --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 64 --no-cpu-baseline --timing-steps 0 > $O/np32_bench_under_rocprof.json 2> $O/np32_prof.err; echo "np32 rocprof rc=$?"
python scripts/prof_summary.py gpurun_out/prof_np/np32_results.db > $O/np32_kernel_stats.csv; head -n 8 $O/np32_kernel_stats.csv | cut -c1-150
find gpurun_out/prof_np -size +20M -delete
bash scripts/configs.sh > $O/configs.log 2>&1; cp gpurun_out/configs.jsonl $O/configs.jsonl
