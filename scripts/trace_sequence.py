#!/usr/bin/env python
"""Durations (us) of the encoder launches of one layer, in launch order, from a rocprofv3 --kernel-trace run of scripts/encode_only.py.
usage: trace_sequence.py <dir> <label> [first=40] [count=9]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
g = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm3", "gemm2", "gemm_kernel", "attn_enc", "layernorm"))]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 40
count = int(sys.argv[4]) if len(sys.argv) > 4 else 9
half = len(g) // 2
print("==", sys.argv[2], "launches", first, "..", first + count, "of the last pass:")
for r in g[half + first: half + first + count]:
    nm = r["Kernel_Name"]
    nm = "gemm3" if "gemm3" in nm else "gemm2" if "gemm2" in nm else "attn" if "attn" in nm else "layernorm"
    print("   %-10s grid %8s  %7.1f us" % (nm, r["Grid_Size_X"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
last = g[-1]
print("   last launch (cross K/V) %.1f us" % ((int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) / 1e3))
