#!/usr/bin/env python3
"""BASELINE.json configs[4] in small: an ensemble of perturbed columns x the full lbl spectrum, run to the
radiative-convective steady state on 1..8 MI355X, one rank per GPU.  Every rank integrates its block of
columns; one all-reduce(max) of the residual per check keeps the ranks on the same iteration, one
all-gather of (Tlayer[20], T_surface) ends the run.

    python examples/equilibrium_ensemble.py --ncol 12500 --tol 0.01 --max-steps 1500      (one GPU's share)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/equilibrium_ensemble.py --ncol 100000 --tol 0.01 --max-steps 1500

    python examples/equilibrium_ensemble.py --ncol 100000 --shard 3/8 --tol 0.01 --max-steps 1500   (rehearsal: the
        block of rank 3 of 8 alone on one GPU, no collective; tools/eq100k_shards.sh runs all eight that way)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        examples/equilibrium_ensemble.py --backend gloo --ncol 16 --nodes 1025      (two ranks on ONE GPU: the
        collectives run on CPU tensors over gloo -- RCCL does not take two ranks on one device; tests do this)

Needs an MI355X per rank (the product has no CPU path).  Spectra are the seeded synthetic stand-in for the
absent lbl.*.asc files (our_first_climate_model_amd.synth_lbl_tables)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=1000, help="columns in the whole ensemble")
    ap.add_argument("--nodes", type=int, default=131072, help="wavelength nodes of the lbl spectrum")
    ap.add_argument("--tol", type=float, default=0.05, help="K per iteration on the adjusted profile, worst column")
    ap.add_argument("--max-steps", type=int, default=1000)
    ap.add_argument("--check-every", type=int, default=10)
    ap.add_argument("--progress", type=int, default=50, help="stderr line every so many iterations (0: none)")
    ap.add_argument("--shard", default=None, metavar="R/W",
                    help="rehearsal without a collective: integrate the block of rank R of W alone in this process")
    ap.add_argument("--save-state", default=None, help="write the [columns][21] result of this process to a .npy file")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend; gloo (collectives on CPU tensors) to run several ranks on one GPU")
    ap.add_argument("--nan-column", type=int, default=None, metavar="C",
                    help="fault injection: column C of the whole ensemble starts with a NaN surface temperature; it "
                         "must keep EVERY rank from reporting convergence (the NaN flag of the all-reduce)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import our_first_climate_model_amd as rcm
    from our_first_climate_model_amd.distributed import gather_state, run_to_equilibrium, shard_bounds, shard_ensemble

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    shard = None
    if a.shard:
        assert "RANK" not in os.environ, "--shard is the one-process rehearsal; under torchrun the ranks are the shards"
        shard = tuple(int(x) for x in a.shard.split("/"))
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    assert torch.cuda.is_available(), "needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(dev)
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    lo, hi = shard_bounds(a.ncol, *shard) if shard else shard_bounds(a.ncol, rank, world)
    atm, _ = rcm.atm_load(os.path.join(ROOT, "tests", "golden", "fpda.lbl.atm"))
    spec = rcm.perturbation_sweep(hi - lo, perturb_rh=True, perturb_ts=True, first=lo, total=a.ncol)
    cols = rcm.build_columns(atm, spec, "lbl")
    if a.nan_column is not None and lo <= a.nan_column < hi:
        cols["T_surface0"][a.nan_column - lo] = float("nan")
    wvl, tabs = rcm.synth_lbl_tables(a.nodes, plevel=cols["plevel"])
    inp = dict(mode="lbl", ncol=hi - lo, cols=cols, solar=rcm.default_solar()["solar_irr"], wvl=wvl, tabs=tabs)
    ens = shard_ensemble(rcm, inp, device=dev, stream=torch.cuda.current_stream().cuda_stream)  # empty block: no handle

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    def progress(done, worst):
        if rank == 0 and a.progress and done % a.progress == 0:
            print(f"[equilibrium] {done} iterations, worst residual {worst:.4f} K/iteration, "
                  f"{time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)

    steps, resid = run_to_equilibrium(ens, a.tol, a.max_steps, a.check_every, dist, progress)
    packed = torch.empty((hi - lo, 21), dtype=torch.float64, device="cuda")
    ens.pack_state(packed.data_ptr())
    ens.sync()
    on_cpu = dist.is_initialized() and a.backend == "gloo"
    full = packed if shard else gather_state(packed.cpu() if on_cpu else packed, a.ncol, world, dist)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if a.save_state and rank == 0:
        np.save(a.save_state, full.cpu().numpy())
    if rank == 0:
        ts = full[:, 20].cpu().numpy()
        what = (f"columns {lo}..{hi - 1} (rank {shard[0]} of {shard[1]}, alone) of {a.ncol}" if shard else f"{a.ncol} columns")
        print(json.dumps({
            "workload": f"{what} x {a.nodes - 1} lbl bands to equilibrium on {world} GPU(s)",
            "steps": steps, "converged": bool(resid <= a.tol), "residual_K_per_step": resid, "tol_K": a.tol,
            "seconds": elapsed, "column_timesteps_per_sec": (hi - lo if shard else a.ncol) * steps / elapsed,
            "T_surface_K": {"min": float(ts.min()), "median": float(np.median(ts)), "max": float(ts.max())},
            "all_finite": bool(torch.isfinite(full).all()), "ranks": world, "backend": a.backend if world > 1 else None}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
