"""Worker of tests/test_gpu_distributed.py::test_two_ranks_shared_parameter_training_*: one rank of a gloo group (two ranks share
the one GPU; on an 8-GPU node the same code runs with backend "nccl" = RCCL) -- or the single process when WORLD_SIZE is unset --
training shared soil parameters on its shard of the basin's columns with the REAL HIP engine (agent.DifferentiableLGAR)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    tmp, out, n_columns, epochs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lgar_py_amd import config
    from lgar_py_amd import workloads as W
    from lgar_py_amd.agent import DifferentiableLGAR
    from test_host_io import write_forcing, write_soil_dat
    f = W.synth1_forcing()
    d = os.path.join(tmp, "w%d_r%d" % (world, rank))
    os.makedirs(d, exist_ok=True)
    soil = write_soil_dat(os.path.join(d, "soil.dat"))
    forcing = write_forcing(os.path.join(d, "f.csv"), f, step_min=5)
    ov = {"data.forcing_file": forcing, "data.soil_params_file": soil, "models.hyperparameters.epochs": epochs,
          "models.hyperparameters.learning_rate": 2e-4, "models.hyperparameters.warmup": 0, "n_columns": n_columns}
    cfg = config.load_config(data="synth_1", models="five_minute", cwd=tmp, overrides=ov)
    scale = 0.5 + np.arange(n_columns) / float(n_columns)  # uneven rainfall over the basin
    agent = DifferentiableLGAR(cfg, observations=0.05 * np.ones(f.shape[0]), log=lambda s: None, forcing_scale=scale)
    agent.run()
    params = torch.cat([p.detach().reshape(-1).double().cpu() for p in agent.model.parameters()]).numpy()
    np.savez(out % rank, params=params, loss=np.array([h["loss"] for h in agent.history]), lo=agent.lo, hi=agent.hi,
             in_sync=agent.parameters_in_sync(), sharded=agent.sharded)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
