"""The mixed-precision mode on the bench ensemble (1M columns x 144 steps, fp64 state, fp32-transcendental trapezoid): three
timed passes, for profiling (tools/pmc_passes.sh workload "mix").  (dev tool)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
P = W.perturbed_columns(N, seed=0)
sc = W.forcing_scale(N, seed=1000)
f = W.synth1_forcing()
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                    ponded_depth_max=0.0, dtype=torch.float64, geff_precision="f32")
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[None, :]).contiguous()
pe = torch.zeros_like(pr)
out = {k: torch.empty_like(pr) for k in ("runoff", "percolation")}
for _ in range(3):
    eng.reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.forward(pr, pe, series=("runoff", "percolation"), out=out, basin=("runoff",), check=False)
    b.record()
    torch.cuda.synchronize()
    print("mixed %d columns: %.2f ms, %.3e column-timesteps/s" % (N, a.elapsed_time(b), N * f.shape[0] / (a.elapsed_time(b) * 1e-3)), flush=True)
