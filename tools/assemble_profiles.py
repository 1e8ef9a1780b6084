#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one profiling session (gpurun_out/<tag>{a,b,f,w,g,f4}, kt<tag>) into the
committed summaries profiles/r01_final_*.  usage: assemble_profiles.py TAG BENCH_JSON"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, bench = sys.argv[1], sys.argv[2]


def agg(d):
    fs = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    a, kn = collections.defaultdict(list), None
    for r in csv.DictReader(open(fs[0])):
        if "k_rt_lbl" in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kn = (r["Kernel_Name"], r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"])
    return {k: sum(v) / len(v) for k, v in a.items()}, kn


a, kn = agg(f"p{tag}a")
b, _ = agg(f"p{tag}b")
f, _ = agg(f"p{tag}f")
w, _ = agg(f"p{tag}w")
g, _ = agg(f"p{tag}g")
f4, _ = agg(f"p{tag}f4")
ks = list(csv.DictReader(open(glob.glob(f"gpurun_out/kt{tag}/*/*kernel_stats.csv")[0])))
rt = [r for r in ks if "k_rt_lbl" in r["Name"]][0]
t = float(rt["AverageNs"]) * 1e-9
bands64 = 10000 * 131071 / 64
clock = g["GRBM_GUI_ACTIVE"] / 8 / t
true4 = 5 * 20 * 131072 * 8 + 131072 * 8
out = {
    "kernel": kn[0], "vgpr_count_reported": kn[1], "lds_block_size": kn[2], "scratch_size": kn[3], "grid_size": kn[4],
    "note": "rocprofv3 --pmc, one pass per counter group, each its own run of `python3 bench.py --steps 2 --warmup 1 "
            "--no-cpu-baseline` (10000 columns x 131071 bands); values are per k_rt_lbl_pair launch (mean of 3)",
    "kernel_time_s_from_kernel_trace": t,
    "counters": {**a, **b, **g},
    "derived": {"valu_instructions_per_64_bands": a["SQ_INSTS_VALU"] / bands64,
                "valu_busy_fraction": a["SQ_ACTIVE_INST_VALU"] * 4 / (t * clock * 1024),
                "clock_GHz_from_GRBM_GUI_ACTIVE": clock / 1e9,
                "l2_hit_rate": g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])},
    "hbm_traffic": {"workload": "bench.py default (10000 columns x 131071 bands)", "FETCH_SIZE_KB": f["FETCH_SIZE"],
                    "WRITE_SIZE_KB": w["WRITE_SIZE"],
                    "calibration": {"run": "bench.py --ncol 4: every table byte is needed once per launch",
                                    "FETCH_SIZE_KB": f4["FETCH_SIZE"], "true_read_bytes": true4,
                                    "ratio": f4["FETCH_SIZE"] * 1024 / true4,
                                    "conclusion": "FETCH_SIZE counts 1/2 of the bytes for this 8-B-per-lane coalesced "
                                                  "pattern too (MI355X_MICROARCH.md HBM section): read bytes = 2 x "
                                                  "FETCH_SIZE x 1024"},
                    "traffic_bytes_per_launch": 2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024,
                    "algorithmic_bytes_per_launch": 808 * 131071 * 10000}}
json.dump(out, open("profiles/r01_final_pmc.json", "w"), indent=1)
shutil.copy(glob.glob(f"gpurun_out/kt{tag}/*/*kernel_stats.csv")[0], "profiles/r01_final_kernel_stats_bench10k.csv")
shutil.copy(bench, "profiles/r01_final_bench_10k.json")
print(json.dumps(out["derived"], indent=1))
print("traffic", out["hbm_traffic"]["traffic_bytes_per_launch"], "kernel s", t)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
          "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
    print(k, round(out["counters"][k] / bands64, 1))
d = json.load(open(bench))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["cpu_baseline"])
