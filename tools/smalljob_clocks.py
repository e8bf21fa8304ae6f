"""Cycle attribution of a one-wave job (one Phillipsburg column x 3000 h, fp64, cooperating lanes): the `clocks` measurement
variant (tools/ablate.py build with LGAR_VARIANTS=clocks) adds the shader-clock cycles between consecutive measurement points
of the grid's first wave to one slot per point (csrc/lgar_measure.hpp LGAR_POINT_CLK).  usage: python tools/smalljob_clocks.py
[LANES] [native|f32] (dev tool)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LGAR_LIB", os.path.join(ROOT, "lgar_py_amd", "csrc", "variants", "liblgar_hip_clocks.so"))
import numpy as np
import torch
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

NAMES = {0: "loop: forcing load, series stores, accumulators, GIUH tail", 1: "free-drainage front, AET", 2: "insert_water",
         3: "sweep: rest (check_column_mass, loop exit)", 4: "event scan / passes", 5: "create front, ponded depth",
         6: "calc_dzdt: loop exit", 7: "mass_balance", 8: "GIUH, NaN check",
         10: "calc_dzdt: find the next moving front", 11: "calc_dzdt: pick, riders' theta(psi), Se", 12: "trapezoid: ends (+ riders)",
         13: "trapezoid: heads", 14: "trapezoid: nodes", 15: "trapezoid: terms", 16: "trapezoid: sum",
         17: "calc_dzdt: dz/dt arithmetic", 19: "sweep: loop control", 20: "sweep: layer-bottom front (psi continuity)",
         22: "sweep: in-layer front, thetas before the search", 23: "sweep: search + theta", 24: "sweep: psi from theta, carry"}
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
precision = sys.argv[2] if len(sys.argv) > 2 else "native"  # "f32": the mixed-precision mode (MODE 6)
g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
f, P, T, N = g["forcing"], W.PHILLIPSBURG, 3000, 1
eng = lg.LgarEngine(*[P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=N, dt_h=1.0,
                    ponded_depth_max=2.0, dtype=torch.float64, forward_lanes=lanes, geff_precision=precision)
pr = torch.tensor(f[:T, 0:1], device="cuda").expand(T, N).contiguous()
pe = torch.tensor(f[:T, 1:2], device="cuda").expand(T, N).contiguous()
buf = (ctypes.c_ulonglong * 64)()
for rep in range(2):
    eng.reset()
    eng.lib.lgar_debug_clocks(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.forward(pr, pe, series=("runoff",), check=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
assert eng.lib.lgar_debug_clocks(buf, 0) == 0
tot = sum(buf[i] for i in range(32))
print(json.dumps(dict(lanes=lanes, geff_precision=precision, ms=round(ms, 2), cycles_total=tot, cycles_per_step=round(tot / T, 1))))
for i in range(32):
    if buf[32 + i]:
        print("%2d %-62s %6.2f %%  %9.1f cycles/step  %8.2f hits/step  %8.1f cycles/hit" % (
            i, NAMES.get(i, "?"), 100.0 * buf[i] / tot, buf[i] / T, buf[32 + i] / T, buf[i] / buf[32 + i]))
