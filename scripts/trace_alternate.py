#!/usr/bin/env python
"""rocprofv3 --kernel-trace of a run in which every gemm2 launch is issued twice (-DWLX_PROBE_GEMM_TWICE): average duration of the FIRST and of the
SECOND launch of each pair, per (kernel, grid). usage: trace_alternate.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
g = [r for r in rows if "gemm2" in r["Kernel_Name"]]
acc = collections.OrderedDict()
for i in range(0, len(g) - 1, 2):
    a, b = g[i], g[i + 1]
    k = (a["Kernel_Name"].split("(")[0][-28:], a["Grid_Size_X"], a["Grid_Size_Y"])
    if (b["Kernel_Name"], b["Grid_Size_X"]) != (a["Kernel_Name"], a["Grid_Size_X"]):
        continue
    e = acc.setdefault(k, [0, 0, 0])
    e[0] += 1; e[1] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"]); e[2] += int(b["End_Timestamp"]) - int(b["Start_Timestamp"])
print("(kernel, grid): pairs, first launch avg us, second launch avg us")
t1 = t2 = 0
for k, (n, a, b) in acc.items():
    print("  ", k, n, round(a / n / 1e3, 2), round(b / n / 1e3, 2)); t1 += a; t2 += b
print("   sums per run (us):", round(t1 / 1e3, 1), round(t2 / 1e3, 1))
