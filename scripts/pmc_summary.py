#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc pass: per kernel name, launches and mean counter value per launch.
usage: pmc_summary.py <rocprof output dir>  (reads *counter_collection.csv)"""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
print("kernel,launches,counter,mean_per_launch,total")
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    for c, v in acc[k].items():
        n = max(1, len(disp[k]))
        print(f'"{k}",{n},{c},{v / n:.1f},{v:.1f}')
