#!/usr/bin/env python
"""Per-(kernel, grid) launch durations of the LAST pass of a rocprofv3 --kernel-trace run of scripts/encode_only.py.
usage: trace_by_grid.py <dir with *kernel_trace.csv> <label> [passes=2]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = list(csv.DictReader(open(f)))
g = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm3", "gemm2", "gemm_kernel", "attn_enc", "layernorm", "prep_window"))]
n = len(g) // passes
acc = collections.OrderedDict()
for r in g[-n:]:
    k = (r["Kernel_Name"].split("(")[0][-36:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    a = acc.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("==", sys.argv[2], "last pass, by (kernel, grid): launches, avg us, total us")
for k, (c, t) in acc.items():
    print("  ", k, c, round(t / c / 1e3, 1), round(t / 1e3, 1))
print("   total us", round(sum(t for c, t in acc.values()) / 1e3, 1))
