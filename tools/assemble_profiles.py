#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of one profiling session (tools/profile_session.sh TAG WORKLOAD: gpurun_out/p<TAG>{a,b,f,
w,g,f4}, kt<TAG>, bench_<TAG>.json) into the committed summaries profiles/<ROUND>_<workload>_{pmc.json,
kernel_stats.csv,bench.json}.  usage: assemble_profiles.py TAG [lbl|repwvl100] [ROUND]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_id, the byte constants)

tag = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "lbl"
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
b = json.load(open(f"gpurun_out/bench_{tag}.json"))
KERNEL = b["roofline"]["kernel"]
PM_STEPS = KT_STEPS = 1
ncol = b["config"]["columns_total"]
nspec = b["config"]["bands"] if wl == "lbl" else b["config"]["wavelengths"]
points = ncol * nspec
per_point = bench.BYTES_PER_BAND if wl == "lbl" else bench.BYTES_PER_RWVL


def agg(d):
    fs = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    a, kn = collections.defaultdict(list), None
    for r in csv.DictReader(open(fs[0])):
        if KERNEL in r["Kernel_Name"]:
            a[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kn = (r["Kernel_Name"], r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"])
    big = {k: max(v) for k, v in a.items()}  # fused kernel: the warm-up launch covers fewer steps; take the timed one
    if PM_STEPS > 1:
        return {k: big[k] / PM_STEPS for k in a}, kn
    return {k: sum(v) / len(v) for k, v in a.items()}, kn


a, kn = agg(f"p{tag}a")
bb, _ = agg(f"p{tag}b")
f, _ = agg(f"p{tag}f")
w, _ = agg(f"p{tag}w")
g, _ = agg(f"p{tag}g")
f4, _ = agg(f"p{tag}f4")
ks_path = glob.glob(f"gpurun_out/kt{tag}/*/*kernel_stats.csv")[0]
ks = list(csv.DictReader(open(ks_path)))
rt = [r for r in ks if KERNEL in r["Name"]][0]
t = float(rt["AverageNs"]) * 1e-9 if KT_STEPS == 1 else float(rt["MaxNs"]) * 1e-9 / KT_STEPS
clock = g["GRBM_GUI_ACTIVE"] / 8 / t
if wl == "lbl":
    true_cal = 5 * 20 * (nspec + 1) * 8 + (nspec + 1) * 8
    cal_run = "bench.py --ncol 4: every table byte is needed once per launch"
else:
    true_cal = None  # one column touches only its (T, p) brackets: no clean calibration; the lbl one holds
    cal_run = "bench.py --workload repwvl100 --ncol 1 (informational)"
out = {
    "kernel": kn[0], "kernel_source_id": bench.kernel_source_id(), "workload": b["config"]["workload"],
    "vgpr_count_reported": kn[1], "lds_block_size": kn[2], "scratch_size": kn[3], "grid_size": kn[4],
    "note": "rocprofv3 --pmc, one pass per counter group, each its own run of bench.py --workload " + wl +
            " --no-cpu-baseline; values are per launch of the kernel (mean over the launches of the run)",
    "kernel_time_s_from_kernel_trace": t,
    "counters": {**a, **bb, **g},
    "derived": {"valu_instructions_per_point": a["SQ_INSTS_VALU"] * 64 / points / 64,
                "valu_instructions_per_64_points": a["SQ_INSTS_VALU"] / (points / 64),
                "valu_busy_fraction": a["SQ_ACTIVE_INST_VALU"] * 4 / (t * clock * 1024),
                "valu_issue_frac_at_2.4GHz": a["SQ_INSTS_VALU"] * 4 / (t * 2.4e9 * 1024),
                "clock_GHz_from_GRBM_GUI_ACTIVE": clock / 1e9,
                "l2_hit_rate": g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])},
    "hbm_traffic": {"workload": b["config"]["workload"], "FETCH_SIZE_KB": f["FETCH_SIZE"],
                    "WRITE_SIZE_KB": w["WRITE_SIZE"],
                    "calibration": {"run": cal_run, "FETCH_SIZE_KB": f4.get("FETCH_SIZE"), "true_read_bytes": true_cal,
                                    "ratio": (f4["FETCH_SIZE"] * 1024 / true_cal) if true_cal else None,
                                    "conclusion": "FETCH_SIZE counts 1/2 of the bytes for this 8-B-per-lane coalesced "
                                                  "pattern (MI355X_MICROARCH.md HBM section): read bytes = 2 x "
                                                  "FETCH_SIZE x 1024"},
                    "traffic_bytes_per_launch": 2 * f["FETCH_SIZE"] * 1024 + w["WRITE_SIZE"] * 1024,
                    "algorithmic_bytes_per_launch": per_point * points}}
base = os.path.join(ROOT, "profiles", f"{rnd}_{wl}")
json.dump(out, open(base + "_pmc.json", "w"), indent=1)
shutil.copy(ks_path, base + "_kernel_stats.csv")
shutil.copy(f"gpurun_out/bench_{tag}.json", base + "_bench.json")
print(json.dumps(out["derived"], indent=1))
print("traffic", out["hbm_traffic"]["traffic_bytes_per_launch"], "kernel s", t)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
          "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
    print(k, "per 64 points", round(out["counters"][k] / (points / 64), 1))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], b.get("cpu_baseline"))
