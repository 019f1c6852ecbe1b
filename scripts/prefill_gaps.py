#!/usr/bin/env python3
"""Prefill timeline from a rocprofv3 kernel-trace CSV: the window between the first and the last prompt GEMM (k_mmq_wide), the time
kernels were running inside it, the idle time by the kernel that FOLLOWS the gap, and the largest single gaps (micro-batch boundaries)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("mi355x::", "").split("(")[0]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_mmq_wide" in r[2]]
w = rows[max(0, idx[0] - 3):idx[-1] + 1]
span = (w[-1][1] - w[0][0]) / 1e3
busy = sum(e - s for s, e, _ in w) / 1e3
print(f"prefill window: {len(w)} kernels, span {span:.1f} us, kernels running {busy:.1f} us, idle {span - busy:.1f} us ({100 * (span - busy) / span:.1f} %)")
by = {}
gaps = []
for i in range(len(w) - 1):
    g = (w[i + 1][0] - w[i][1]) / 1e3
    a = by.setdefault(w[i + 1][2][:60], [0, 0.0]); a[0] += 1; a[1] += g
    gaps.append((g, w[i][2][:40], w[i + 1][2][:40]))
for k, (n, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  idle before {k:60s}: {n:4d} x {g / n:7.2f} us = {g:8.1f} us")
for g, a, b in sorted(gaps, reverse=True)[:10]:
    print(f"  gap {g:8.1f} us between {a} and {b}")
kt = {}
for s, e, k in w:
    a = kt.setdefault(k[:60], [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for k, (n, t) in sorted(kt.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k:60s}: {n:4d} x {t / n:7.2f} us = {t:8.1f} us")
