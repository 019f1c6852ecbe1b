#!/usr/bin/env python3
"""Decode-step timeline from a rocprofv3 kernel-trace CSV: per step (launches between two output-matrix mat-vecs) the span, the
time kernels were running, the idle gaps inside the step and the idle time between steps."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if ("T_Q6K, false, 2>" in r[2] or "T_Q6KP, false, 2>" in r[2] or "T_Q6KS, false, 2>" in r[2])]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    st = rows[a + 1:b + 1]
    if len(st) < 50: continue
    busy = sum(e - s for s, e, _ in st)
    inner = sum(max(0, st[i + 1][0] - st[i][1]) for i in range(len(st) - 1))
    steps.append((st[0][0], st[-1][1], busy, inner, len(st)))
tail = steps[-16:]
for i, (s, e, busy, inner, n) in enumerate(tail):
    nxt = tail[i + 1][0] - e if i + 1 < len(tail) else 0
    print(f"step kernels={n} span={(e - s) / 1e3:8.1f} us busy={busy / 1e3:8.1f} inner_gaps={inner / 1e3:6.1f} gap_to_next_step={nxt / 1e3:6.1f}")
# The host's turn between two steps: a step ends with the output mat-vec, then the logits' D2H copy (a blit kernel when the
# target is pinned memory), then NOTHING runs until the host has synchronised, sampled, staged the next inputs, recognised the graph and called
# hipGraphLaunch.  The cut above puts that idle time inside the next step's "inner gaps"; here it is on its own.
turn = []
for a in ends[-17:-1]:
    nxt = rows[a + 1:a + 4]
    if not nxt: continue
    out_end = rows[a][1]
    if "copyBuffer" in nxt[0][2] and len(nxt) > 1:
        turn.append(((nxt[0][0] - out_end) / 1e3, (nxt[0][1] - nxt[0][0]) / 1e3, (nxt[1][0] - nxt[0][1]) / 1e3, nxt[1][2].split("(")[0][:40]))
    else:
        turn.append(((nxt[0][0] - out_end) / 1e3, 0.0, 0.0, nxt[0][2].split("(")[0][:40]))
if turn:
    n = len(turn)
    print(f"between steps (mean of {n}): output mat-vec end -> D2H copy start {sum(t[0] for t in turn) / n:6.1f} us, copy {sum(t[1] for t in turn) / n:5.1f} us, "
          f"copy end -> first kernel of the next step ({turn[-1][3]}) {sum(t[2] for t in turn) / n:6.1f} us  [host turnaround: sync + sampler + inputs + graph key + hipGraphLaunch]")
if "-v" in sys.argv:  # every launch of one step: start offset, duration, gap before it
    stv = rows[ends[-3] + 1:ends[-2] + 1]
    t0 = stv[0][0]
    for i, (s_, e_, k_) in enumerate(stv):
        g_ = (s_ - stv[i - 1][1]) / 1e3 if i else 0.0
        print(f"  {(s_ - t0) / 1e3:9.2f} us  dur {(e_ - s_) / 1e3:7.2f}  gap {g_:6.2f}  {k_.replace('mi355x::', '').split('(')[0][:70]}")
big = {}
st = rows[ends[-3] + 1:ends[-2] + 1]
for i in range(len(st) - 1):
    g = (st[i + 1][0] - st[i][1]) / 1e3
    k = st[i + 1][2].replace("mi355x::", "").split("(")[0][:50]
    a = big.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += g
for k, (n, g) in sorted(big.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  gap before {k:50s}: {n:3d} x {g / n:5.2f} us")
# round 6: WHERE the sporadic inner gaps sit — every gap > 1.5 us of the last 8 steps with its index in the step and the kernels either side
if "-w" in sys.argv:
    for si in range(-10, -2):
        stw = rows[ends[si] + 1:ends[si + 1] + 1]
        out = []
        for i in range(1, len(stw)):
            g_ = (stw[i][0] - stw[i - 1][1]) / 1e3
            if g_ > 1.5:
                out.append(f"#{i}:{g_:.1f}us[{stw[i - 1][2].replace('mi355x::', '').split('(')[0][5:28]}->{stw[i][2].replace('mi355x::', '').split('(')[0][5:28]}]")
        print(f"  step {si}: " + " ".join(out))
