"""Multi-GPU plumbing: columns are independent (no horizontal coupling anywhere in main.cpp), so the
ensemble is cut into contiguous blocks, one per rank, and the only exchange is ONE all-gather of
(Tlayer[20], T_surface) per column at the end.  torch.distributed is the
transport ('nccl' = RCCL over xGMI on the MI355X node, 'gloo' in the CPU tests)."""
import numpy as np


def shard_bounds(ncol_total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; every rank but the last gets ceil(N/world)."""
    per = (ncol_total + world - 1) // world
    lo = min(rank * per, ncol_total)
    return lo, min(lo + per, ncol_total)


class EmptyShard:
    """The rank that holds NO columns (9 columns on 8 ranks: ranks 5..7).  rcm_create refuses an ensemble of zero
    columns, so such a rank creates no handle and touches no kernel -- but it must still join every collective
    (the residual all-reduce of run_to_equilibrium, the final all-gather, the barriers), or the other ranks hang.
    Same surface as Ensemble as far as the multi-rank drivers use it; every call is a no-op."""
    ncol = 0

    def step(self, n=1):
        pass

    def get_residual(self):
        return np.zeros(0)

    def pack_state(self, dev_ptr):
        pass

    def sync(self):
        pass

    def set_rt_timing(self, on):
        pass

    def get_rt_timing(self):
        return 0.0, 0

    def set_spectral_chunks(self, n):
        pass

    def get_spectral_chunks(self):
        return 0

    def close(self):
        pass


def shard_ensemble(pkg, inp, **kw):
    """configure(Ensemble(ncol_local, ...), inp) for a rank's block of columns; EmptyShard when the block is empty."""
    if inp["ncol"] == 0:
        return EmptyShard()
    return pkg.configure(pkg.Ensemble(inp["ncol"], **kw), inp)


def gather_state(local_packed, ncol_total, world, dist=None):
    """All-gathers the packed [ncol_local][nlev] state blocks into [ncol_total][nlev] on every rank.
    `local_packed` is a torch tensor on the rank's device (or CPU for gloo).  Shards are padded to
    ceil(N/world) rows so that one equal-size all-gather does it."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_packed
    per = (ncol_total + world - 1) // world
    nlev = local_packed.shape[1]
    send = local_packed
    if send.shape[0] < per:
        pad = torch.zeros((per - send.shape[0], nlev), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    recv = torch.empty((world * per, nlev), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    return recv[:ncol_total]


def run_to_equilibrium(ens, tol_K, max_steps, check_every=10, dist=None, progress=None):
    """Steps every rank's shard `check_every` iterations at a time until the WORST column of the whole
    ensemble has a residual <= tol_K (rcm_get_residual: change of the adjusted profile per iteration)
    or max_steps have run: one all-reduce(max) of one double per check, so all ranks stop after the
    same number of iterations and the gathered state is one consistent iteration of the ensemble
    (BASELINE configs[4]).  Returns (steps_done, worst_residual_K).
    `ens` needs step(n) and get_residual() (our_first_climate_model_amd.Ensemble); `progress(done, worst)` is called
    after every check.  A NaN residual anywhere (a column gone NaN) never counts as converged."""
    import torch
    if dist is None:
        import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    done, worst = 0, float("inf")
    while done < max_steps:
        n = min(check_every, max_steps - done)
        ens.step(n)
        done += n
        r = ens.get_residual()
        worst = (float("nan") if np.isnan(r).any() else float(np.max(r))) if len(r) else 0.0  # an empty shard never holds the others back
        if multi:
            backend = dist.get_backend()
            # (max over ranks is not NaN-propagating in every backend: the NaN travels as a flag)
            nan = worst != worst
            t = torch.tensor([0.0 if nan else worst, 1.0 if nan else 0.0], dtype=torch.float64,
                             device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float("nan") if float(t[1]) > 0 else float(t[0])
        if progress is not None:
            progress(done, worst)
        if worst <= tol_K:
            break
    return done, worst
