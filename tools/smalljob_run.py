"""One small fp64 job for profiling: N replicated Phillipsburg columns x 3000 h with a given lanes-per-column setting.
usage: python tools/smalljob_run.py N LANES [reps]   (dev tool)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

N, lanes = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
f, P, T = g["forcing"], W.PHILLIPSBURG, 3000
eng = lg.LgarEngine(*[P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=N, dt_h=1.0,
                    ponded_depth_max=2.0, dtype=torch.float64, forward_lanes=lanes)
pr = torch.tensor(f[:T, 0:1], device="cuda").expand(T, N).contiguous()
pe = torch.tensor(f[:T, 1:2], device="cuda").expand(T, N).contiguous()
for _ in range(reps):
    eng.reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.forward(pr, pe, series=("runoff",), check=False)
    b.record()
    torch.cuda.synchronize()
    print("N %d lanes %d: %.2f ms, geff wave calls %d" % (N, lanes, a.elapsed_time(b), eng.geff_wave_calls()), flush=True)
