#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (name, calls, total_us, avg_us, pct): what --stats prints, from the trace itself."""
import csv, sys
agg = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        a = agg.setdefault(r["Kernel_Name"].replace("mi355x::", ""), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values()) or 1
print("kernel,calls,total_us,avg_us,pct")
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.3f,%.3f,%.2f' % (k, n, ns / 1e3, ns / 1e3 / n, 100.0 * ns / tot))
