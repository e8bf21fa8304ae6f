"""What the in-kernel basin aggregation costs: the forward launch of the synth_1 ensemble with and without basin=("runoff",).
(dev tool)  usage: python tools/basin_cost.py [f32|f64] [N]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

dt = torch.float32 if (sys.argv[1] if len(sys.argv) > 1 else "f32") == "f32" else torch.float64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
P = W.perturbed_columns(N, seed=0)
sc = torch.tensor(W.forcing_scale(N, seed=1000), device="cuda")
f = W.synth1_forcing()
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                    ponded_depth_max=0.0, dtype=dt)
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * sc[None, :]).to(dt).contiguous()
pe = torch.zeros_like(pr)
out = {k: torch.empty_like(pr) for k in ("runoff", "percolation")}
for label, kw in (("no_basin", {}), ("basin_runoff", {"basin": ("runoff",)})) * 2:
    ms = []
    for _ in range(6):
        eng.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.forward(pr, pe, series=("runoff", "percolation"), out=out, check=False, **kw)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms = sorted(ms[1:])
    print(json.dumps({"mode": label, "ms_median": ms[len(ms) // 2], "ms_min": ms[0], "faulted": int((eng.status != 0).sum())}), flush=True)
