"""Multi-GPU: columns shard embarrassingly, one process per GPU.

Each rank owns the contiguous column range workloads.shard_bounds(N, world, rank): parameters, forcing and state
are sharded identically, no halo, no data-path collective.  The ONE exchange of the path is the basin-runoff
reduction: an all-reduce(SUM) of the per-timestep runoff vector [T] after the time loop (RCCL over xGMI with the
"nccl" backend; 576 B .. 24 KB, latency-bound).  The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as dist

from .workloads import shard_bounds


def _collective_on(group=None):
    """A process group exists and the exchange has someone to talk to -- or LGAR_FORCE_DIST=1 asks for the collective to run
    even in a group of one (how a one-GPU box executes the RCCL path of configs[3]: bench.py, tests/test_gpu_distributed.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("LGAR_FORCE_DIST", "0") == "1"


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_reduce(t, group=None):
    """all-reduce(SUM) in place.  RCCL ("nccl") reduces device tensors over xGMI; with the gloo backend (CPU tests, and
    the rehearsal of several ranks on one GPU) a device tensor goes through the host."""
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def basin_runoff(local_series, weights=None, group=None):
    """local_series: [T, n_local] per-step runoff of this rank's columns (any float dtype, any device).
    Returns the basin total per timestep [T] in fp64, summed over every rank's columns
    (optionally area-weighted with weights [n_local])."""
    s = local_series.to(torch.float64)
    if weights is not None:
        s = s * weights.to(torch.float64)[None, :]
    total = s.sum(dim=1)
    if _collective_on(group):
        _all_reduce(total, group)
    return total


def all_reduce_sum(t, group=None):
    """In-place all-reduce(SUM) of an already locally reduced tensor (e.g. the kernel's in-epilogue basin sums [T])."""
    if _collective_on(group):
        _all_reduce(t, group)
    return t


def reduce_parameter_gradients(grads, group=None):
    """Shared-parameter training: all-reduce(SUM) of the [L x 3] gradient scalars."""
    if _collective_on(group):
        flat = torch.cat([g.reshape(-1) for g in grads])
        _all_reduce(flat, group)
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].reshape(g.shape))
            o += g.numel()
    return grads


class ShardedColumns:
    """This rank's shard of an N-column job.

    engine_factory(params_shard: dict name -> [L, n_local] numpy, **engine_kw) builds the compute engine; the
    default is the HIP engine on this rank's GPU (tests inject a CPU checker to exercise the sharding and the
    collective under gloo)."""

    def __init__(self, params, n_total, rank=None, world=None, engine_factory=None, **engine_kw):
        r, w = world_info()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.n_total = n_total
        self.lo, self.hi = shard_bounds(n_total, self.world, self.rank)
        shard = {k: v[:, self.lo:self.hi] for k, v in params.items()}
        if engine_factory is None:
            from .engine import LgarEngine

            def engine_factory(p, **kw):
                return LgarEngine(p["alpha"], p["n"], p["ksat"], p["theta_e"], p["theta_r"], p["thickness"], **kw)
        self.engine = engine_factory(shard, **engine_kw)

    def shard(self, x):
        """Slice a [T, N_total] (or [N_total]) array to this rank's columns."""
        return x[..., self.lo:self.hi]

    def run(self, precip, pet, **kw):
        """precip/pet: this rank's [T, n_local] forcing.  Returns (local per-step series dict, basin runoff [T])."""
        out = self.engine.forward(precip, pet, **kw)
        return out, basin_runoff(out["runoff"])
