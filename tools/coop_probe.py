"""Cooperating lanes (LgarDims.forward_lanes) on small jobs: timing of BASELINE configs[1] (10 000 replicated Phillipsburg columns
x 3000 h, fp64) and of one column, per lanes-per-column setting; every setting must reproduce forward_lanes=1 bit for bit.
(dev tool)  usage: python tools/coop_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
f = g["forcing"]
P = W.PHILLIPSBURG


def run(N, lanes, T=3000, reps=2):
    eng = lg.LgarEngine(*[P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=N, dt_h=1.0,
                        ponded_depth_max=2.0, dtype=torch.float64, forward_lanes=lanes)
    pr = torch.tensor(f[:T, 0:1], device="cuda").expand(T, N).contiguous()
    pe = torch.tensor(f[:T, 1:2], device="cuda").expand(T, N).contiguous()
    best = 1e9
    for _ in range(reps):
        eng.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = eng.forward(pr, pe, series=("runoff", "percolation", "infiltration", "AET"), check=False)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, out, eng


for N in (10000, 1, 64, 1000):
    base = None
    for lanes in (1, 4, 6, 8, 16, 64, 0):
        if N * max(lanes, 1) > 64 * 4096:
            continue
        ms, out, eng = run(N, lanes)
        sig = (out["infiltration"].clone(), out["AET"].clone(), out["runoff"].clone(), eng.depth.clone(), eng.theta.clone(),
               eng.psi.clone(), eng.n_fronts.clone(), eng.totals.clone())
        same = None
        if base is None:
            base = sig
        else:
            same = all(torch.equal(x, y) for x, y in zip(sig, base))
        print(json.dumps(dict(columns=N, forward_lanes=lanes, first_cap=os.environ.get("LGAR_COOP_FIRST_CAP", "8"), ms=round(ms, 2),
                              col_steps_per_s=N * 3000 / (ms * 1e-3), bitwise_equal_to_one_lane=same,
                              faulted=int((eng.status != 0).sum()))), flush=True)
