#!/bin/bash
# Round 4: everything profiles/r04_* holds except the parity log (copied from the pytest run): default bench line, rocprofv3 kernel stats
# of the same command, PMC passes for the decode / prefill kernels AND for the -np 32 configuration, the secondary configs with rooflines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/collect; rm -rf $O; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log | cut -c1-400
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1200 $O/bench.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --no-cpu-baseline --pmc-traffic 0 > $O/bench_under_rocprof.json 2> $O/prof.err; echo "rocprof rc=$?"
python scripts/prof_summary.py gpurun_out/prof/bench_results.db > $O/bench_kernel_stats.csv; head -n 14 $O/bench_kernel_stats.csv | cut -c1-150
find gpurun_out/prof -size +20M -delete
PMC_GROUPS="FETCH_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  BENCH_ARGS="--steps 16 --warmup 2 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0" TOPN=40 bash scripts/prof_pmc.sh > $O/pmc_passes.txt 2>&1
for i in 0 1 2; do cp gpurun_out/pmc_$i.summary.txt $O/pmc_pass$i.csv 2>/dev/null; done
# -np 32 (config 3): kernel stats + its own PMC passes (none existed)
rm -rf gpurun_out/prof_np
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_np -o np32 -- python bench.py --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 64 --no-cpu-baseline --timing-steps 0 > $O/np32_bench_under_rocprof.json 2> $O/np32_prof.err; echo "np32 rocprof rc=$?"
python scripts/prof_summary.py gpurun_out/prof_np/np32_results.db > $O/np32_kernel_stats.csv; head -n 10 $O/np32_kernel_stats.csv | cut -c1-150
find gpurun_out/prof_np -size +20M -delete
PMC_GROUPS="FETCH_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  BENCH_ARGS="--preset llama3-8b-q4_k_m --np 32 --prefill 64 --steps 16 --warmup 2 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0" TOPN=24 bash scripts/prof_pmc.sh > $O/np32_pmc_passes.txt 2>&1
for i in 0 1 2; do cp gpurun_out/pmc_$i.summary.txt $O/np32_pmc_pass$i.csv 2>/dev/null; done
head -n 5 $O/np32_pmc_pass0.csv | cut -c1-200
CONFIGS_PMC=0 bash scripts/configs.sh > $O/configs.log 2>&1
for cfg in "--np 32 --draft 4" "--np 32 --draft 1" "--np 1 --draft 8" "--np 1 --draft 16"; do
  timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 --prefill 128 --steps 64 $cfg 2>> gpurun_out/configs.err >> gpurun_out/configs.jsonl
done
cp gpurun_out/configs.jsonl $O/configs.jsonl; wc -l $O/configs.jsonl
python3 - <<'PY'
import json
for ln in open("gpurun_out/collect/configs.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    r = d.get("roofline") or {}
    print("%-60s %9.1f %s  %.3f ms  prefill %s  roofline %s %.3f (%s us)" % (d["config"]["workload"][:60], d["value"], d["unit"], d["ms_per_step"], d.get("prefill_tok_s"), r.get("kernel"), r.get("frac") or 0, r.get("avg_us")))
PY
