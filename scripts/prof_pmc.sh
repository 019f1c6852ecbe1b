#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass, kernel-trace only) over a SHORT bench run; per-kernel averages.
# usage: PMC_GROUPS="A B C;D E" BENCH_ARGS="..." bash scripts/prof_pmc.sh   (groups separated by ';')
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
GROUPS_="${PMC_GROUPS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM}"
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --prefill 512 --timing-steps 0 --no-cpu-baseline --layers 4}"
i=0
IFS=';' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  rm -rf gpurun_out/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d gpurun_out/pmc_$i -o pmc --output-format csv -- python bench.py $ARGS > gpurun_out/pmc_$i.json 2> gpurun_out/pmc_$i.err
  echo "== pass $i: $g (rc=$?)"
  f=$(find gpurun_out/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" ${PMC_FILTER:-} | tee gpurun_out/pmc_$i.summary.txt | head -${TOPN:-12} | cut -c1-260; else tail -5 gpurun_out/pmc_$i.err; fi
  find gpurun_out/pmc_$i -size +8M -delete
  i=$((i+1))
done
