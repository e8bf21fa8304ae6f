"""Worker of tests/test_gpu_distributed.py: one rank of a gloo group (or, LGAR_TEST_BACKEND=nccl, of an RCCL group); every rank drives the REAL HIP engine on the (shared)
GPU through ShardedColumns and writes its shard's outputs to an .npz file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(N):
    from lgar_py_amd import workloads as W
    P = W.perturbed_columns(N, seed=3)
    sc = W.forcing_scale(N, 0.5, 1.0, seed=4)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    return P, pr, np.zeros_like(pr)


def main():
    N, out = int(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("LGAR_TEST_BACKEND", "gloo")
    grouped = world > 1 or backend == "nccl"
    if backend == "nccl":
        # RCCL group (of one on a one-GPU box; LGAR_FORCE_DIST=1 makes the exchange run all the same): before any other GPU call
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lgar_py_amd.distributed import ShardedColumns
    P, pr, pe = problem(N)
    sh = ShardedColumns(P, N, dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float64, device="cuda:0")
    res, basin = sh.run(torch.tensor(sh.shard(pr)), torch.tensor(sh.shard(pe)), check=False)
    np.savez(out % rank, lo=sh.lo, hi=sh.hi, runoff=res["runoff"].cpu().numpy(), basin=basin.cpu().numpy(),
             status=sh.engine.status.cpu().numpy())
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
