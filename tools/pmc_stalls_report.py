#!/usr/bin/env python3
"""Prints the counters of tools/pmc_stalls.sh for one kernel, per launch (max over launches: the timed one of a
fused multi-step kernel) and relative to wave-cycles.  usage: pmc_stalls_report.py TAG KERNEL_SUBSTRING"""
import collections
import csv
import glob
import sys

tag, kern = sys.argv[1], sys.argv[2]
vals = {}
for f in sorted(glob.glob(f"gpurun_out/st{tag}_*/*/*counter_collection.csv")):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in a.items():
        vals[k] = max(v)
ks = glob.glob(f"gpurun_out/st{tag}_kt/*/*kernel_stats.csv")
if ks:
    for r in csv.DictReader(open(ks[0])):
        if kern in r["Name"]:
            print("kernel", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "max ns", r["MaxNs"])
wc = vals.get("SQ_WAVE_CYCLES", 1.0)
for k in sorted(vals):
    print(f"{k:40s} {vals[k]:14.5g}   / wave-cycles {vals[k] / wc:8.4f}")
