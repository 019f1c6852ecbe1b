#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv): one line per kernel, counters as columns."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name") or row.get("kernel_name")
        if flt and flt not in k:
            continue
        c = row.get("Counter_Name") or row.get("counter_name")
        v = float(row.get("Counter_Value") or row.get("counter_value") or 0)
        acc[k][c] += v
        disp[k].add(row.get("Dispatch_Id") or row.get("dispatch_id"))
names = sorted({c for k in acc for c in acc[k]})
print("kernel,dispatches," + ",".join(names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = max(1, len(disp[k]))
    print(f"\"{k[:60]}\",{n}," + ",".join(f"{acc[k][c] / n:.4g}" for c in names))
