#!/usr/bin/env python3
"""Timeline of ONE decode step out of a rocprofv3 results db: per kernel start offset, duration and the idle gap before it.
usage: decode_timeline.py <bench_results.db> [step_from_end]   (a step = the launches between two output-matrix mat-vecs)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the output projection (Q6_K stream kernel with the norm prologue) ends every decode step
ends = [i for i, r in enumerate(rows) if "k_mmvq_stream<mi355x::T_Q6K, false, 2>" in r[0] or "k_mmvq_stream<T_Q6K, false, 2>" in r[0]]
if len(ends) < back + 1:
    sys.exit("not enough decode steps in the trace")
lo, hi = ends[-back - 1] + 1, ends[-back]
step = rows[lo:hi + 1]
t0 = step[0][1]
busy = gaps = 0
prev_end = None
print("start_us,dur_us,gap_before_us,kernel")
agg = {}
for name, s, e in step:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    dur = (e - s) / 1e3
    busy += dur
    gaps += max(gap, 0.0)
    short = name.replace("mi355x::", "").split("(")[0][:60]
    a = agg.setdefault(short, [0, 0.0, 0.0])
    a[0] += 1; a[1] += dur; a[2] += max(gap, 0.0)
    prev_end = e
    if "-v" in sys.argv:
        print(f"{(s - t0) / 1e3:.2f},{dur:.2f},{gap:.2f},{short}")
print(f"# kernels={len(step)} span_us={(step[-1][2] - t0) / 1e3:.1f} busy_us={busy:.1f} gaps_us={gaps:.1f}")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"# {n:4d} x {d / n:7.2f} us  (+{g / n:5.2f} us gap before)  {k}")
