#!/bin/bash
# PMC pass over the shadow-plane GEMM probe: LDS conflicts / stalls / instruction mix of the gate/up-shaped launch (grid 448 or 896)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
BIN=${1:-scripts/ubench/shadow_probe_f1.bin}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/shpmc; rocprofv3 --kernel-trace --pmc $grp -d /tmp/shpmc -o p --output-format csv -- $BIN > /dev/null 2>&1
  f=$(find /tmp/shpmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "k_mmq_shadow" in r["Kernel_Name"] and r["Grid_Size"] in (str(448*512), str(896*256), str(448*512)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
