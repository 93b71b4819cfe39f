"""How do N concurrent slots (one HIP stream + captured decode-step graph each) share the GPU? Reads a rocprofv3
`--kernel-trace` CSV of `bench.py --streams N` and prints: the hardware queues the dispatches went to (Queue_Id) and which
HIP streams / host threads fed each, per-queue busy time, the time the device had 0 / 1 / 2 / ... kernels in flight, and how
often consecutive dispatches of ONE queue belong to different streams (streams multiplexed onto one hardware queue serialise
on its in-order barrier packets).   usage: python scripts/stream_overlap.py <kernel_trace.csv> [decode-only]"""
import csv
import sys
from collections import Counter, defaultdict

path = sys.argv[1]
decode_only = len(sys.argv) > 2
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"]
        if decode_only and not (nm.startswith("dec_") or nm.startswith("search_") or "dec_" in nm):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), int(r["Stream_Id"]), int(r["Thread_Id"]), nm))
rows.sort()
print(f"{len(rows)} dispatches, span {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms")
if decode_only:
    # steady state only: the middle of the decode dispatches (set-up, warm-up and the tail are left out)
    lo, hi = rows[int(len(rows) * 0.45)][0], rows[int(len(rows) * 0.95)][0]
    rows = [r for r in rows if lo <= r[0] <= hi]
    print(f"steady-state window: {len(rows)} dispatches, {(hi - lo) / 1e6:.2f} ms")
byq = defaultdict(list)
for s, e, q, st, th, nm in rows:
    byq[q].append((s, e, st, th))
for q, v in sorted(byq.items()):
    streams = Counter(x[2] for x in v)
    threads = Counter(x[3] for x in v)
    busy = sum(e - s for s, e, _, _ in v)
    switches = sum(1 for a, b in zip(v, v[1:]) if a[2] != b[2])
    print(f"queue {q}: {len(v)} dispatches, busy {busy / 1e6:.2f} ms, streams {dict(streams)}, host threads {len(threads)} {dict(threads)}, "
          f"stream switches between consecutive dispatches {switches}")
# concurrency histogram
ev = []
for s, e, *_ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], Counter()
for t, d in ev:
    hist[depth] += t - last
    depth += d; last = t
tot = sum(hist.values())
print("kernels in flight -> share of the span:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per stream: dispatch-to-dispatch period (start to next start) for the decode chain
bys = defaultdict(list)
for s, e, q, st, th, nm in rows:
    bys[st].append((s, e))
for st, v in sorted(bys.items()):
    if len(v) < 100:
        continue
    gaps = sorted(b[0] - a[1] for a, b in zip(v, v[1:]) if b[0] - a[1] < 50000)
    durs = sorted(e - s for s, e in v)
    print(f"stream {st}: {len(v)} dispatches, median kernel {durs[len(durs) // 2] / 1e3:.2f} us, median gap to next {gaps[len(gaps) // 2] / 1e3:.2f} us, "
          f"p90 gap {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us")
