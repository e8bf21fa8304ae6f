"""CPU, world_size 2 (gloo): shared-parameter training with the basin's columns sharded over the ranks
(lgar_py_amd.agent.DifferentiableLGAR under a process group; reference loop: agents/DifferentiableLGAR.py:94-172, SURVEY.md
section 8e: "training with shared parameters adds an all-reduce of L x 3 gradient scalars").  The compute engine is injected:
the device code compiled for the host (tests/devsim behind tests/_sim_lgar_engine.SimLgarEngine) stands in for the HIP engine,
so model, autograd tape, tangent launches and the agent's two exchanges all run here; on GPUs the same code runs with backend
"nccl" (= RCCL; tests/test_gpu_distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, TESTS

N_COLUMNS, EPOCHS = 5, 2


def _agent(tmp, world_rank=None):
    sys.path[:0] = [ROOT, TESTS]
    import _sim_lgar_engine
    _sim_lgar_engine.install()
    from lgar_py_amd import config
    from lgar_py_amd import workloads as W
    from lgar_py_amd.agent import DifferentiableLGAR
    from test_host_io import write_forcing, write_soil_dat
    f = W.synth1_forcing()[:96]
    tag = "single" if world_rank is None else "r%d" % world_rank
    os.makedirs(os.path.join(tmp, tag), exist_ok=True)
    soil = write_soil_dat(os.path.join(tmp, tag, "soil.dat"))
    forcing = write_forcing(os.path.join(tmp, tag, "f.csv"), f, step_min=5)
    ov = {"data.forcing_file": forcing, "data.soil_params_file": soil, "models.hyperparameters.epochs": EPOCHS,
          "models.hyperparameters.learning_rate": 2e-4, "models.hyperparameters.warmup": 0, "models.endtime": 8.0,
          "n_columns": N_COLUMNS, "device": "cuda:0"}
    cfg = config.load_config(data="synth_1", models="five_minute", cwd=tmp, overrides=ov)
    scale = 0.6 + 0.2 * np.arange(N_COLUMNS)  # uneven rainfall: the columns differ, and so do the ranks' shares of the gradient
    obs = 0.05 * np.ones(f.shape[0])
    agent = DifferentiableLGAR(cfg, observations=obs, log=lambda s: None, forcing_scale=scale)
    return agent


def _params(agent):
    return torch.cat([p.detach().reshape(-1).double() for p in agent.model.parameters()]).numpy()


def _worker(rank, world, port, tmp, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    agent = _agent(tmp, rank)
    assert agent.sharded and (agent.rank, agent.world) == (rank, world)
    calls = []
    from lgar_py_amd import distributed as D
    real = D.reduce_parameter_gradients
    D.reduce_parameter_gradients = lambda grads, group=None: (calls.append(len(grads)), real(grads, group))[1]
    agent.run()
    in_sync = agent.parameters_in_sync()
    q.put((rank, agent.lo, agent.hi, agent.model.n_columns, _params(agent), [h["loss"] for h in agent.history], in_sync, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_shared_parameter_training_equals_single_process(tmp_path):
    single = _agent(str(tmp_path))
    assert not single.sharded and single.model.n_columns == N_COLUMNS
    p0 = _params(single).copy()
    single.run()
    p1 = _params(single)
    assert np.abs(p1 - p0).max() > 1e-4  # the parameters moved (Adam: ~ the learning rate per epoch)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2], r[3]) for r in res] == [(0, 3, 3), (3, 5, 2)]          # contiguous shards of 3 and 2 columns
    assert all(r[6] for r in res)                                               # bit-equal parameters on every rank
    assert np.array_equal(res[0][4].view(np.int64), res[1][4].view(np.int64))
    assert all(r[7] == [9] * EPOCHS for r in res)                               # L x 3 gradients reduced once per epoch
    for r in res:
        assert np.abs(r[4] - p1).max() <= 1e-12, np.abs(r[4] - p1).max()        # ... equal to the single-process run
        assert np.allclose(r[5], [h["loss"] for h in single.history], rtol=1e-12, atol=0)
