"""Per-(kernel, grid) average durations from a rocprofv3 kernel trace CSV (decode-step kernels of different shapes share a name)."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][-60:]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if pat in k[0]:
        print(f"{k[0]:60s} grid={k[1]:>8s} wg={k[2]:>5s} n={len(v):6d} avg={sum(v)/len(v)/1e3:8.2f} us  min={min(v)/1e3:8.2f}")
