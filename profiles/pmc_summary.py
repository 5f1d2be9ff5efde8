#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter.
usage: python profiles/pmc_summary.py [--kernel SUBSTR] gpurun_out/pmc*/*/p_counter_collection.csv"""
import collections
import csv
import sys

args = sys.argv[1:]
kernel_filter = "render"
if "--kernel" in args:
    i = args.index("--kernel")
    kernel_filter = args[i + 1]
    del args[i:i + 2]
tot = collections.OrderedDict()
for f in args:
    for row in csv.DictReader(open(f)):
        if kernel_filter not in row["Kernel_Name"]:
            continue
        key = row["Counter_Name"]
        tot.setdefault(key, [0.0, 0])
        tot[key][0] += float(row["Counter_Value"])
        tot[key][1] += 1
for k, (v, n) in tot.items():
    print(f"{k:34s} {v:20.0f}  (dispatches {n})")
