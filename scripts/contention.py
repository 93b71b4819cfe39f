#!/usr/bin/env python
"""What do N concurrent slots contend on? Two rocprofv3 --kernel-trace CSVs of the same workload (bench.py --streams 1 / --streams N):
per decode kernel the mean DURATION in each, and per stream the mean GAP between the end of a graph node and the start of the next one of
the same stream (the in-graph launch boundary as the command processor delivers it). Execution-resource contention (CUs, LDS, L2, HBM)
shows up in the durations; dispatch contention (command processor, hardware queues) in the gaps.
usage: contention.py <trace_1.csv> <trace_N.csv>"""
import csv
import sys
from collections import defaultdict


def load(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            nm = r["Kernel_Name"]
            if "dec_" in nm or "search_" in nm:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), nm.split("(")[0].replace("void wlx::", "").replace("wlx::", "")[:52]))
    rows.sort()
    lo, hi = rows[int(len(rows) * 0.3)][0], rows[int(len(rows) * 0.95)][0]       # steady state
    rows = [r for r in rows if lo <= r[0] <= hi]
    dur = defaultdict(list)
    for s, e, st, nm in rows:
        dur[nm].append(e - s)
    gaps = []
    by = defaultdict(list)
    for r in rows:
        by[r[2]].append(r)
    for st, v in by.items():
        for a, b in zip(v, v[1:]):
            g = b[0] - a[1]
            if 0 <= g < 20000:                 # inside a graph (step-to-step host gaps are longer)
                gaps.append(g)
    span = (rows[-1][1] - rows[0][0]) / 1e3
    busy = sum(e - s for s, e, _, _ in rows) / 1e3
    return dur, gaps, len(by), span, busy


d1, g1, n1, span1, busy1 = load(sys.argv[1])
dn, gn, nn, spann, busyn = load(sys.argv[2])
print(f"streams: {n1} vs {nn}; mean in-graph gap between consecutive nodes of a stream: {sum(g1) / len(g1):.0f} ns vs {sum(gn) / len(gn):.0f} ns "
      f"(median {sorted(g1)[len(g1) // 2]} vs {sorted(gn)[len(gn) // 2]})")
print(f"sum of kernel durations / wall span of the steady-state window: {busy1 / span1:.2f} vs {busyn / spann:.2f} kernels in flight on average")
print(f"{'kernel':54s} {'calls':>7s} {'1 stream us':>12s} {'N streams us':>13s} {'ratio':>6s}")
tot1 = totn = 0.0
for nm in sorted(d1, key=lambda k: -sum(d1[k])):
    if nm not in dn:
        continue
    a, b = sum(d1[nm]) / len(d1[nm]) / 1e3, sum(dn[nm]) / len(dn[nm]) / 1e3
    per_step = len(d1[nm])
    tot1 += sum(d1[nm]) / 1e3
    totn += sum(dn[nm]) / 1e3 / nn
    print(f"{nm:54s} {len(d1[nm]):7d} {a:12.2f} {b:13.2f} {b / a:6.2f}")
print(f"kernel time per stream-window-equivalent: x{totn / tot1:.2f}")
