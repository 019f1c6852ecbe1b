#!/bin/bash
# Round 6: everything profiles/r06_* holds except the parity log (copied from the pytest run) and the A/B files: smoke, the default bench line (with its
# parity object), rocprofv3 kernel stats of the same command, PMC passes for the decode / prefill kernels and for -np 32, the replayed-step timeline
# with the host's turn on its own, the secondary configurations.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/collect; rm -rf $O; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log | cut -c1-400
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench --output-format csv -- python bench.py --no-cpu-baseline --pmc-traffic 0 > $O/bench_under_rocprof.json 2> $O/prof.err; echo "rocprof rc=$?"
f=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_stats.py $f > $O/bench_kernel_stats.csv; head -n 14 $O/bench_kernel_stats.csv | cut -c1-150
# (the replayed-step timeline needs a trace WITHOUT the eager per-launch timing pass at its end: decode_gaps.py reads the last 16 steps)
rm -rf gpurun_out/prof_gaps
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_gaps -o bench --output-format csv -- python bench.py --no-cpu-baseline --pmc-traffic 0 --timing-steps 0 > /dev/null 2>> $O/prof.err
python scripts/decode_gaps.py $(find gpurun_out/prof_gaps -name "*kernel_trace.csv" | head -1) > $O/decode_gaps.txt; tail -12 $O/decode_gaps.txt
rm -rf gpurun_out/prof_gaps
find gpurun_out/prof -size +8M -delete
PMC_GROUPS="FETCH_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA;SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
  BENCH_ARGS="--steps 16 --warmup 2 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0" TOPN=40 bash scripts/prof_pmc.sh > $O/pmc_passes.txt 2>&1
for i in 0 1 2; do cp gpurun_out/pmc_$i.summary.txt $O/pmc_pass$i.csv 2>/dev/null; done
head -n 6 $O/pmc_pass0.csv | cut -c1-200
rm -rf gpurun_out/prof_np
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_np -o np32 --output-format csv -- python bench.py --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 64 --no-cpu-baseline --timing-steps 0 > $O/np32_bench_under_rocprof.json 2> $O/np32_prof.err; echo "np32 rocprof rc=$?"
f=$(find gpurun_out/prof_np -name "*kernel_trace.csv" | head -1)
python scripts/trace_stats.py $f > $O/np32_kernel_stats.csv; head -n 10 $O/np32_kernel_stats.csv | cut -c1-150
find gpurun_out/prof_np -size +8M -delete
PMC_GROUPS="FETCH_SIZE" BENCH_ARGS="--preset llama3-8b-q4_k_m --np 32 --prefill 64 --steps 16 --warmup 2 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0" TOPN=24 bash scripts/prof_pmc.sh > $O/np32_pmc_passes.txt 2>&1
cp gpurun_out/pmc_0.summary.txt $O/np32_pmc_pass0.csv 2>/dev/null; head -n 5 $O/np32_pmc_pass0.csv | cut -c1-200
rm -f gpurun_out/configs.jsonl gpurun_out/configs.err
for cfg in "--np 32 --prefill 128 --steps 64" "--np 8 --prefill 128 --steps 64" "--fa 0" "--fa 0 --np 32 --prefill 128 --steps 64" "--preset qwen2-7b-q5_k_m --prefill 8064 --steps 64" "--preset llama3-70b-q4_k_m --prefill 512 --steps 32" \
           "--preset tinyllama-1.1b-q8_0 --prefill 512" "--np 32 --prefill 128 --steps 64 --ctkv q8_0" "--prefill 7936 --steps 64 --ctkv q8_0" "--prefill 7936 --steps 64 --ctkv q4_0" "--prefill 7936 --steps 64" "--np 32 --draft 4 --prefill 128 --steps 64" "--np 1 --draft 8 --prefill 128 --steps 64" "--emulate-tp 8 --preset llama3-70b-q4_k_m --prefill 512 --steps 32" \
           "--preset tinyllama-1.1b-q8_0 --np 32 --prefill 128 --steps 64" "--preset llama3-8b-q8_0 --np 32 --prefill 128 --steps 32" "--preset llama3-8b-q8_0 --prefill 2048 --steps 64" "--np 2 --prefill 128 --steps 64" "--preset llama2-7b-q4_k_m --prefill 2048 --steps 64" "--preset llama3.2-3b-q4_k_m --prefill 2048 --steps 64" "--preset llama3.2-1b-q4_k_m --prefill 2048 --steps 64"; do
  timeout 900 python bench.py --cpu-steps 8 --pmc-traffic 0 --timing-steps 8 $cfg 2>> gpurun_out/configs.err >> gpurun_out/configs.jsonl
done
cp gpurun_out/configs.jsonl $O/configs.jsonl; wc -l $O/configs.jsonl
python3 - <<'PY'
import json
for ln in open("gpurun_out/collect/configs.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    r = d.get("roofline") or {}
    par = d.get("parity") or {}
    b1 = par.get("batch_1_rows", par) or {}
    cb = par.get("continuous_batch") or {}
    print("%-78s %9.1f %s  %.3f ms  prefill %s  roofline %s %.3f (%s us) | parity batch-1 within_bar=%s ids %s rel %.2e%s" % (d["config"]["workload"][:78], d["value"], d["unit"], d["ms_per_step"], d.get("prefill_tok_s"), r.get("kernel"), r.get("frac") or 0, r.get("avg_us"),
          b1.get("within_bar"), b1.get("argmax_agree"), b1.get("max_abs_rel_to_logit_range") or 0, (" | batch within_bar=%s ids %s rel %.2e" % (cb.get("within_bar"), cb.get("argmax_agree"), cb.get("max_abs_rel_to_logit_range") or 0)) if cb else ""))
PY
