"""CPU, world_size 2 (gloo): the column sharding and the basin-runoff all-reduce of lgar_py_amd.distributed.
The compute engine is injected: here the CPU oracle stands in for the HIP engine (tests may use the oracle as the
checker); on GPUs the same code path runs with backend "nccl" (= RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class OracleEngine:
    def __init__(self, p, dt_h, pdm):
        self.p, self.dt_h, self.pdm = p, dt_h, pdm

    def forward(self, precip, pet, **kw):
        from oracle import lgar_oracle as O
        ro, pc, acc, st = O.run_columns(self.p["alpha"], self.p["n"], self.p["ksat"], self.p["theta_e"],
                                        self.p["theta_r"], self.p["thickness"], precip, pet, pdm=self.pdm,
                                        dt_h=self.dt_h, nthreads=2)
        return {"runoff": torch.tensor(ro), "percolation": torch.tensor(pc)}


def _factory(p, **kw):
    return OracleEngine(p, kw["dt_h"], kw["ponded_depth_max"])


def _problem(N):
    from lgar_py_amd import workloads as W
    P = W.perturbed_columns(N, seed=3)
    sc = W.forcing_scale(N, 0.5, 1.0, seed=4)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    return P, pr, np.zeros_like(pr)


def _worker(rank, world, port, N, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lgar_py_amd.distributed import ShardedColumns, reduce_parameter_gradients
    P, pr, pe = _problem(N)
    sh = ShardedColumns(P, N, engine_factory=_factory, dt_h=300.0 / 3600.0, ponded_depth_max=0.0)
    assert (sh.rank, sh.world) == (rank, world)
    out, basin = sh.run(sh.shard(pr), sh.shard(pe))
    g = [torch.full((3,), float(rank + 1), dtype=torch.float64) for _ in range(3)]
    reduce_parameter_gradients(g)
    q.put((rank, sh.lo, sh.hi, out["runoff"].numpy(), basin.numpy(), g[0].numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_single_process():
    from lgar_py_amd.distributed import ShardedColumns
    N, world = 37, 2
    P, pr, pe = _problem(N)
    single = ShardedColumns(P, N, rank=0, world=1, engine_factory=_factory, dt_h=300.0 / 3600.0, ponded_depth_max=0.0)
    out1, basin1 = single.run(pr, pe)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == N
    ro = np.concatenate([r[3] for r in res], axis=1)
    assert np.array_equal(ro, out1["runoff"].numpy())          # identical per-column outputs, however sharded
    for r in res:
        assert np.allclose(r[4], basin1.numpy(), rtol=1e-12, atol=1e-12)  # same reduced runoff up to summation order
        assert np.array_equal(r[5], np.full(3, 3.0))           # 1 + 2: gradient all-reduce
    assert basin1.sum() > 0


@pytest.mark.timeout(300)
def test_eight_rank_ragged_sharding_matches_single_process():
    """configs[3]'s world size on the CPU (gloo): 8 ranks over a column count that 8 does not divide (shards of 6 and 5
    columns), so the last ranks' bounds, the concatenation order and the [T] all-reduce are those an 8-GPU node will see."""
    import socket
    from lgar_py_amd.distributed import ShardedColumns
    from lgar_py_amd.workloads import shard_bounds
    N, world = 43, 8
    P, pr, pe = _problem(N)
    single = ShardedColumns(P, N, rank=0, world=1, engine_factory=_factory, dt_h=300.0 / 3600.0, ponded_depth_max=0.0)
    out1, basin1 = single.run(pr, pe)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [shard_bounds(N, world, k) for k in range(world)]
    assert sorted(r[2] - r[1] for r in res) == [5] * 5 + [6] * 3
    ro = np.concatenate([r[3] for r in res], axis=1)
    assert np.array_equal(ro, out1["runoff"].numpy())
    for r in res:
        assert np.allclose(r[4], basin1.numpy(), rtol=1e-12, atol=1e-12)
        assert np.array_equal(r[5], np.full(3, 36.0))          # 1 + 2 + ... + 8


def _force_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lgar_py_amd import distributed as D
    calls = []
    real = D._all_reduce
    D._all_reduce = lambda t, group=None: (calls.append(1), real(t, group))[1]
    s = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    os.environ.pop("LGAR_FORCE_DIST", None)
    a = D.basin_runoff(s)
    n_plain = len(calls)
    os.environ["LGAR_FORCE_DIST"] = "1"
    b = D.basin_runoff(s)
    q.put((n_plain, len(calls), a.numpy(), b.numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_force_dist_runs_the_collective_in_a_group_of_one():
    """LGAR_FORCE_DIST=1: the basin all-reduce executes even when the group has one member (how a one-GPU box exercises the RCCL
    path, tests/test_gpu_distributed.py); without it a group of one skips the exchange.  Same result either way."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(0, 1, port, q))
    p.start()
    n_plain, n_total, a, b = q.get(timeout=100)
    p.join(30)
    assert n_plain == 0 and n_total == 1 and np.array_equal(a, b) and np.array_equal(a, np.arange(12.0).reshape(4, 3).sum(1))
