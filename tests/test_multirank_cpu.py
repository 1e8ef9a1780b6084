"""CPU, world_size 2 and 8, gloo: the column sharding and the single end-of-run all-gather.
The compute inside each rank is the CPU oracle here (no GPU in this container); on the MI355X node
bench.py runs the same shard/gather code with the HIP ensemble and the nccl (= RCCL) backend.
World size 8 is the target node's: 10 000 columns (equal shards), 10 001 (uneven) and 9 (ranks 5..7 hold NO columns:
they create no ensemble -- distributed.EmptyShard -- and still join every collective)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, ncol_total, q, steps=3):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import helpers as H
    import oracle_lib as ol
    import our_first_climate_model_amd as pkg
    from our_first_climate_model_amd.distributed import EmptyShard, gather_state, shard_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(ncol_total, rank, world)
        atm_cols, _ = pkg.atm_load(os.path.join(ROOT, "tests", "golden", "test.atm"))
        spec = pkg.perturbation_sweep(hi - lo, perturb_rh=True, perturb_ts=True, first=lo, total=ncol_total)
        cols = pkg.build_columns(atm_cols, spec, "repwvl")
        inp = dict(mode="repwvl", ncol=hi - lo, cols=cols, solar=pkg.default_solar()["solar_irr"],
                   table=ol.read_rtab(os.path.join(ROOT, "tests", "golden", "Reduced10Forcing.rtab")))
        if hi == lo:                       # the rank without columns: no ensemble at all (as bench.py: shard_ensemble)
            e = EmptyShard()
            e.step(steps)
            packed = torch.zeros((0, 21), dtype=torch.float64)
        else:
            e = H.oracle_ensemble(ol.load(), inp, threads=1)
            e.step(steps)
            s = e.get_state()
            packed = torch.from_numpy(np.concatenate([s["Tlayer"], s["T_surface"][:, None]], axis=1))
        full = gather_state(packed, ncol_total, world, dist)
        if rank == 0:
            q.put(full.numpy().copy())
        else:
            assert full.shape == (ncol_total, 21)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ncol_total,steps", [(2, 7, 3), (2, 8, 3), (8, 9, 3), (8, 10000, 1), (8, 10001, 1)])
def test_ranks_shard_and_gather_equals_one_rank(rcm, olib, world, ncol_total, steps):
    """Shard, step, ONE all-gather -- against the whole ensemble on one rank, bit for bit.  World size 8 with 9 columns
    leaves ranks 5..7 without columns (no ensemble is created there; they pad the all-gather with zeros); 10 001
    columns make the last shard one column short of the others' 1 251."""
    import torch.multiprocessing as mp
    import helpers as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, ncol_total, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=400)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one rank, whole ensemble
    inp = H.make_inputs(rcm, "repwvl", ncol_total, table="Reduced10")
    e = H.oracle_ensemble(olib, inp, threads=8)
    e.step(steps)
    s = e.get_state()
    want = np.concatenate([s["Tlayer"], s["T_surface"][:, None]], axis=1)
    np.testing.assert_array_equal(full, want)  # same CPU arithmetic per column: bit-identical


def test_shard_bounds_cover_the_ensemble_once(rcm):
    from our_first_climate_model_amd.distributed import shard_bounds
    for n in (1, 7, 8, 9, 10000, 10001, 100003):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert max(h - l for l, h in edges) == (n + world - 1) // world
            assert all(h >= l for l, h in edges)
    assert [shard_bounds(9, r, 8) for r in (4, 5, 7)] == [(8, 9), (9, 9), (9, 9)]   # empty shards exist: EmptyShard


def test_sweep_windows_concatenate_to_the_full_sweep(rcm):
    full = rcm.perturbation_sweep(11, perturb_rh=True, perturb_ts=True)
    parts = [rcm.perturbation_sweep(hi - lo, perturb_rh=True, perturb_ts=True, first=lo, total=11)
             for lo, hi in ((0, 6), (6, 11))]
    for k in ("co2_factor", "rh_scale", "T_surface0"):
        np.testing.assert_array_equal(np.concatenate([getattr(p, k) for p in parts]), getattr(full, k))
    s0 = full.shard(1, 2)
    np.testing.assert_array_equal(s0.co2_factor, parts[1].co2_factor)


def _rank_equilibrium(rank, world, port, ncol_total, q, nan_column=None, tol=0.5, max_steps=60):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import helpers as H
    import oracle_lib as ol
    import our_first_climate_model_amd as pkg
    from our_first_climate_model_amd.distributed import EmptyShard, run_to_equilibrium, shard_bounds
    import torch
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(ncol_total, rank, world)
        atm_cols, _ = pkg.atm_load(os.path.join(ROOT, "tests", "golden", "test.atm"))
        spec = pkg.perturbation_sweep(hi - lo, perturb_rh=True, perturb_ts=True, first=lo, total=ncol_total)
        cols = pkg.build_columns(atm_cols, spec, "repwvl")
        if nan_column is not None and lo <= nan_column < hi:
            cols["T_surface0"][nan_column - lo] = float("nan")
        inp = dict(mode="repwvl", ncol=hi - lo, cols=cols, solar=pkg.default_solar()["solar_irr"],
                   table=ol.read_rtab(os.path.join(ROOT, "tests", "golden", "Reduced10Forcing.rtab")))
        e = EmptyShard() if hi == lo else H.oracle_ensemble(ol.load(), inp, threads=1)
        done, worst = run_to_equilibrium(e, tol, max_steps, check_every=5, dist=dist)
        r = e.get_residual()
        q.put((rank, done, worst, float(r.max()) if len(r) else None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ncol_total", [(2, 6), (8, 9), (8, 20)])
def test_ranks_stop_on_the_global_residual(rcm, olib, world, ncol_total):
    """Every rank runs until the worst column of the WHOLE ensemble is below the tolerance: same step count on all,
    equal to the one-rank run of the full ensemble.  8 ranks x 9 columns: ranks 5..7 hold nothing, report residual 0
    to the all-reduce and stop with the others."""
    import torch.multiprocessing as mp
    import helpers as H
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_equilibrium, args=(r, world, port, ncol_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    inp = H.make_inputs(rcm, "repwvl", ncol_total, table="Reduced10")
    e = H.oracle_ensemble(olib, inp, threads=2)
    done1, worst1 = e.run_to_equilibrium(0.5, 60, check_every=5)
    assert 5 <= done1 < 60  # the criterion, not the step cap, ended the run
    assert [g[1] for g in got] == [done1] * world
    assert all(g[2] == worst1 for g in got)
    local = [g[3] for g in got if g[3] is not None]
    assert len(local) == sum(1 for r in range(world) if r * ((ncol_total + world - 1) // world) < ncol_total)
    assert max(local) == worst1  # the global value is the largest of the local ones


def test_a_nan_column_on_one_rank_keeps_every_rank_from_converging(rcm):
    """The all-reduce(max) of the residual carries a NaN as a FLAG (max is not NaN-propagating in every backend): a
    column gone NaN on rank 1 must hold rank 0 -- whose own columns pass any tolerance -- to the step cap as well."""
    import math
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_equilibrium, args=(r, 2, port, 6, q, 5, 1e9, 15)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))  # (rank, steps, global worst, local worst)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [15, 15] and all(math.isnan(g[2]) for g in got)
    assert math.isfinite(got[0][3]) and math.isnan(got[1][3])   # rank 0's own residuals are finite, rank 1 holds the NaN


@pytest.mark.parametrize("ngpus", [2, 8])
def test_bench_self_launch_reaches_every_rank_without_a_gpu(ngpus):
    """`python bench.py --gpus N` with no launcher in front must start torch.distributed.run itself (VERDICT r02
    item 2 / ADVICE r02): on this GPU-less container all N ranks come up (RANK/WORLD_SIZE set by the child
    launcher) and then fail LOUDLY at the product's "no CPU fallback" assertion -- rc != 0, no JSON line.  The
    same command on the MI355X box is tests/test_gpu_fullsize.py::test_bench_gpus_2_without_a_launcher_starts_its_own."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test covers the real run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ngpus), "--backend", "gloo", "--ncol",
                        "9", "--nodes", "257", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and f"--nproc-per-node={ngpus}" in r.stderr
    assert r.stderr.count("bench.py needs the MI355X (no CPU fallback)") >= ngpus, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
