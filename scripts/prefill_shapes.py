#!/usr/bin/env python3
"""Per-layer kernel sequence of the LAST micro-batch of a traced prefill (median duration by position in the layer)."""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("mi355x::", "").split("(")[0][:44]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_mmq_wide" in r[2]]
w = rows[idx[0]:idx[-1] + 1]
# layers start at the norm kernel that precedes the QKV GEMM: split the window at k_rope_qk_store and look back
norms = [i for i, r in enumerate(w) if "k_rope_qk_store" in r[2]]
per = defaultdict(list)
for a, b in zip(norms[-33:-1], norms[-32:]):
    seq = w[a:b]
    key = tuple(k for _, _, k in seq)
    per[key].append([(e - s) / 1e3 for s, e, _ in seq])
for key, lst in sorted(per.items(), key=lambda kv: -len(kv[1]))[:2]:
    print(f"-- {len(lst)} layers with this sequence; total {sum(sorted(x)[len(x)//2] for x in zip(*lst)):.1f} us")
    for j, k in enumerate(key):
        v = sorted(x[j] for x in lst)
        print(f"   {k:46s} {v[len(v) // 2]:8.2f}")
