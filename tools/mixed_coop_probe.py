"""The mixed-precision mode (geff_precision="f32") on small jobs: one lane per column against the library's choice of cooperating
lanes (MODE 6), N replicated Phillipsburg columns x 3000 h; every line must be bit for bit the same results.  (dev tool)
usage: python tools/mixed_coop_probe.py [N ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
f, P, T = g["forcing"], W.PHILLIPSBURG, 3000
for N in [int(x) for x in sys.argv[1:]] or [1, 1000, 5000, 10000, 16384]:
    res = {}
    for lanes in (1, 0):
        eng = lg.LgarEngine(*[P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=N, dt_h=1.0,
                            ponded_depth_max=2.0, dtype=torch.float64, geff_precision="f32", forward_lanes=lanes)
        pr = torch.tensor(f[:T, 0:1], device="cuda").expand(T, N).contiguous()
        pe = torch.tensor(f[:T, 1:2], device="cuda").expand(T, N).contiguous()
        best = 1e9
        for _ in range(3):
            eng.reset()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = eng.forward(pr, pe, series=("runoff", "infiltration", "AET"), check=False)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        res[lanes] = (best, out, eng.totals.clone(), eng.theta.clone(), eng.psi.clone(), eng.dzdt.clone(), eng.cooperating_lanes())
    same = all(torch.equal(res[1][1][k], res[0][1][k]) for k in res[1][1]) and all(torch.equal(res[1][j], res[0][j]) for j in (2, 3, 4, 5))
    print(json.dumps(dict(columns=N, mixed_one_lane_ms=round(res[1][0], 2), mixed_cooperating_ms=round(res[0][0], 2), lanes=res[0][6],
                          bitwise_equal=same)), flush=True)
