#!/usr/bin/env python
"""(kernel, grid) table of the LAST pass of a rocprofv3 --kernel-trace run: everything from the last launch of a marker kernel on.
usage: trace_table.py <dir with *kernel_trace.csv> <label> [marker substring = logmel_kernel] [skip substring,...]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[3] if len(sys.argv) > 3 else "logmel_kernel"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if marker in r["Kernel_Name"])
g = rows[last:]
acc = collections.OrderedDict()
for r in g:
    nm = r["Kernel_Name"].split("(")[0].replace("void wlx::", "").replace("wlx::", "")
    k = (nm[-44:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    a = acc.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = (int(g[-1]["End_Timestamp"]) - int(g[0]["Start_Timestamp"])) / 1e3
print("==", sys.argv[2], "last pass:", len(g), "launches, span", round(span, 1), "us; by (kernel, grid xyz, block): launches, avg us, total us")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  ", k, c, round(t / c / 1e3, 2), round(t / 1e3, 1))
print("   total kernel us", round(sum(t for c, t in acc.values()) / 1e3, 1))
