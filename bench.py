#!/usr/bin/env python3
"""Headline benchmark: column-timesteps/s of the radiative-convective column integrator on the
10 000-column ensemble with the full (synthetic, 131 072-node) lbl.arts 5-species spectrum
(BASELINE.json configs[3]; the real lbl.*.asc files are absent from the reference tree).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the time-loop body main.cpp:531-583 over every column of the ensemble.
The ensemble is sharded by columns over the ranks (strong scaling: the total stays 10 000); the only
exchange is one all-gather of (Tlayer, T_surface) after the K steps, inside the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_BAND = 5 * 20 * 8 + 8  # 5 species x 20 layers x f64 + the wavelength


def host_cores():
    """CPU threads this job may really use: the affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 hardware threads but grants one GPU's share of them)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(pkg, inp_full, n_nodes, log):
    """The CPU oracle (a port of the reference; main.cpp itself needs netcdf and cannot be built here)
    timed on this host's cores on a bounded sample of the same workload."""
    import numpy as np
    import helpers as H
    import oracle_lib as ol
    olib = ol.load()
    cores = host_cores()
    ncol = min(6 * cores, inp_full["ncol"])  # ~13 s of CPU work on the 16 threads of a one-GPU box
    inp = dict(inp_full)
    inp["ncol"] = ncol
    inp["cols"] = {k: (v[:ncol] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == inp_full["ncol"] else v)
                   for k, v in inp_full["cols"].items()}
    e = H.oracle_ensemble(olib, inp, threads=cores)
    t0 = time.perf_counter()
    e.step(1)
    dt = time.perf_counter() - t0
    used = e.threads_used()
    log(f"cpu_baseline: {ncol} columns x 1 step on {used} threads in {dt:.2f} s")
    # the reference itself is single-threaded: one column on one core as well
    inp1 = dict(inp)
    inp1["ncol"] = 1
    inp1["cols"] = {k: (v[:1] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == ncol else v) for k, v in inp["cols"].items()}
    e1 = H.oracle_ensemble(olib, inp1, threads=1)
    t0 = time.perf_counter()
    e1.step(1)
    dt1 = time.perf_counter() - t0
    log(f"cpu_baseline: 1 column x 1 step on 1 thread in {dt1:.2f} s")
    return {"value": ncol / dt, "unit": "column-timesteps/s", "cores": used, "kind": "port",
            "sample": f"{ncol} columns x 1 step x {n_nodes - 1} bands, OpenMP over columns, gcc -O2",
            "single_core_value": 1.0 / dt1}, e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ncol", type=int, default=10000, help="columns in the whole ensemble")
    ap.add_argument("--nodes", type=int, default=131072, help="wavelength nodes of the lbl spectrum")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--chunks", type=int, default=0, help="spectral chunks per column (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend; gloo only to rehearse the N>1 path on a one-GPU box")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: libraries that print banners to fd 1 (RCCL does when its
    # communicator comes up) are sent to stderr until the result is ready
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on the 8-GPU node
    torch.cuda.set_device(dev_index)
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    import our_first_climate_model_amd as pkg
    from our_first_climate_model_amd.distributed import gather_state, shard_bounds
    import helpers as H

    ncol_total = args.ncol * (world if args.scaling == "weak" else 1)
    lo, hi = shard_bounds(ncol_total, rank, world)
    ncol_local = hi - lo

    t_setup = time.perf_counter()
    atm_cols, _ = pkg.atm_load(os.path.join(ROOT, "tests", "golden", "fpda.lbl.atm"))
    spec = pkg.perturbation_sweep(ncol_local, perturb_rh=True, perturb_ts=True, first=lo, total=ncol_total)
    cols = pkg.build_columns(atm_cols, spec, "lbl")
    wvl, tabs = pkg.synth_lbl_tables(args.nodes, plevel=cols["plevel"])
    inp = dict(mode="lbl", ncol=ncol_local, cols=cols, solar=pkg.default_solar()["solar_irr"], wvl=wvl, tabs=tabs)
    stream = torch.cuda.current_stream().cuda_stream
    ens = H.configure(pkg.Ensemble(ncol_local, device=dev_index, stream=stream), inp)
    if args.chunks:
        ens.set_spectral_chunks(args.chunks)
    packed = torch.empty((ncol_local, 21), dtype=torch.float64, device="cuda")
    log(f"setup {time.perf_counter() - t_setup:.1f} s: {ncol_total} columns ({ncol_local} on this rank), "
        f"{args.nodes - 1} bands, {ens.get_spectral_chunks()} spectral chunks/column")

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather():
        # The handle runs on its own HIP stream unless given a non-default one (torch's default stream
        # has handle 0 = "create your own" in the ABI), so order the pack kernel before the collective.
        ens.pack_state(packed.data_ptr())
        ens.sync()
        if not use_dist:
            return packed
        return gather_state(packed if coll_dev == "cuda" else packed.cpu(), ncol_total, world, dist)

    ens.step(args.warmup)
    if use_dist:  # warm the collective too
        gather()
    barrier()
    ens.set_rt_timing(True)
    t0 = time.perf_counter()
    ens.step(args.steps)
    final = gather()
    barrier()
    elapsed = time.perf_counter() - t0
    rt_ms, rt_n = ens.get_rt_timing()
    ens.set_rt_timing(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert final.shape == (ncol_total, 21) and bool(torch.isfinite(final).all())

    if rank == 0:
        nband = args.nodes - 1
        rt_avg_s = rt_ms / 1e3 / max(rt_n, 1)
        alg_bytes = BYTES_PER_BAND * nband * ncol_local  # per RT launch on this rank
        achieved = alg_bytes / rt_avg_s / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE,
        # MI355X_MICROARCH.md HBM section); only valid for the workload they were collected on
        traffic, traffic_src, valu = None, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc.json")))
            prof = pmc["hbm_traffic"]
            if prof["algorithmic_bytes_per_launch"] == alg_bytes:
                traffic, traffic_src = prof["traffic_bytes_per_launch"], "profiles/r01_final_pmc.json"
            # the bound that actually holds: every wave64 VALU instruction occupies its SIMD for 4 cycles;
            # instruction count per 64 bands from the committed PMC pass, time from this run's HIP events
            ipb = pmc["derived"]["valu_instructions_per_64_bands"]
            valu = {"instructions_per_64_bands": ipb, "source": "profiles/r01_final_pmc.json",
                    "issue_frac_at_2.4GHz": ipb * (nband * ncol_local / 64.0) * 4 / (1024 * 2.4e9 * rt_avg_s)}
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "column_timesteps_per_sec", "value": ncol_total * args.steps / elapsed,
            "unit": "column-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{ncol_total}-column ensemble x full lbl.arts 5-species spectrum "
                                   "(BASELINE.json configs[3] at the default 10000); spectra synthetic, seeded",
                       "columns_total": ncol_total, "columns_per_gpu": ncol_local, "lbl_nodes": args.nodes,
                       "bands": nband, "layers": 20, "angles": 30, "atmosphere": "fpda.lbl.atm",
                       "perturbations": "CO2 x linspace(0.5,4), RH x (1+0.1u), T_surface 288.2+2u",
                       "parallelism": f"columns sharded over {world} GPU(s), one all-gather at the end"},
            "roofline": {"bound": "hbm", "kernel": "k_rt_lbl_pair", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": rt_avg_s * 1e3,
                         "launches": rt_n, "fp64_valu_issue": valu,
                         "note": "numerator = 808 B per (column, band), as if every column streamed the 5 species "
                                 "tables itself (SURVEY 8d); the tables are shared, real HBM traffic is ~1% of "
                                 "that and the kernel is fp64-VALU-issue bound (94% VALU busy, profiles/README.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], oe = cpu_baseline(pkg, inp, args.nodes, log)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
