#!/usr/bin/env python3
"""Reads what tools/pmc_compare.sh left under gpurun_out/: per-launch counter means of the RT kernel, product vs variant.
usage: python tools/pmc_compare_report.py VARIANT [kernel-substring=k_rt_lbl_pair]"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = sys.argv[1]
kn = sys.argv[2] if len(sys.argv) > 2 else "k_rt_lbl_pair"
out = {}
for which in ("product", v):
    acc = defaultdict(list)
    for grp in "abg":
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pc_{which}_{grp}", "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for row in csv.DictReader(open(f)):
                if kn in row["Kernel_Name"]:
                    per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (d, c), val in per.items():
                acc[c].append(val)
    out[which] = {c: sum(x) / len(x) for c, x in acc.items()}
print(f"{'counter':28s} {'product':>16s} {v:>16s}  ratio")
for c in sorted(out["product"]):
    a, b = out["product"][c], out[v].get(c, float('nan'))
    print(f"{c:28s} {a:16.4g} {b:16.4g}  {a / b if b else float('nan'):.4f}")
