"""Cycle attribution of the tangent kernel (BASELINE configs[4]: 100 000-column ensemble, forward + backward, fp64): the
`tan_clocks` measurement variant (LGAR_VARIANTS=tan_clocks python tools/ablate.py build) adds the shader-clock cycles between
consecutive measurement points of the grid's FIRST wave (a persistent wave: ~14 blocks of 7 columns x 9 directions x 144 steps)
to one slot per point.  usage: python tools/tangent_clocks.py [N]  (dev tool)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LGAR_LIB", os.path.join(ROOT, "lgar_py_amd", "csrc", "variants", "liblgar_hip_tan_clocks.so"))
import torch
from lgar_py_amd import _capi
from lgar_py_amd import workloads as W
from lgar_py_amd.autograd import lgar_series

NAMES = {0: "step loop: forcing, weights, accumulators, GIUH tail", 1: "free-drainage front, AET", 2: "insert_water (with its Geff)",
         3: "sweep: rest (check_column_mass, loop exit)", 4: "event scan / passes", 5: "create front (with its Geff), ponded depth",
         6: "calc_dzdt: loop exit", 7: "mass_balance", 8: "GIUH, NaN check", 10: "calc_dzdt: find the next moving front",
         17: "calc_dzdt: K(theta), bottom sum, dz/dt (after the trapezoid)", 19: "sweep: loop control",
         20: "sweep: layer-bottom front (psi continuity)", 22: "sweep: in-layer front, thetas before the search",
         23: "sweep: search + theta", 24: "sweep: psi from theta, carry", 25: "Geff: Se, entry", 26: "Geff: the four end evaluations",
         27: "Geff: dh, safe-node count", 28: "Geff: shared blocks (nodes split over the column's lanes)", 29: "Geff: remaining nodes, result"}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dtype = torch.float64
E = W.ensemble_columns(N, seed=0)
f = W.synth1_forcing(); T = f.shape[0]
pr = torch.tensor(f[:, 0:1], device="cuda").expand(T, N).contiguous().to(dtype); pe = torch.zeros_like(pr)
P = {k: torch.tensor(v, device="cuda", dtype=dtype) for k, v in E.items()}
for k in ("alpha", "n", "ksat"):
    P[k].requires_grad_(True)
lib = _capi.load()
buf = (ctypes.c_ulonglong * 64)()
for rep in range(2):
    for k in ("alpha", "n", "ksat"):
        P[k].grad = None
    st = []
    runoff, _ = lgar_series(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                            dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=dtype, check=False, status_out=st)
    loss = torch.mean(runoff[:, st[0] == 0] ** 2)
    torch.cuda.synchronize()
    lib.lgar_debug_clocks_tangent(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); loss.backward(); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
assert lib.lgar_debug_clocks_tangent(buf, 0) == 0
tot = sum(buf[i] for i in range(32))
steps = buf[32 + 0] or 1  # hits of the step loop's point = wave-steps of the first wave
print(json.dumps(dict(columns=N, backward_ms=round(ms, 2), cycles_total=tot, wave_steps=steps, cycles_per_wave_step=round(tot / steps, 1))))
for i in range(32):
    if buf[32 + i]:
        print("%2d %-62s %6.2f %%  %9.1f cycles/step  %8.2f hits/step  %8.1f cycles/hit" % (
            i, NAMES.get(i, "?"), 100.0 * buf[i] / tot, buf[i] / steps, buf[32 + i] / steps, buf[i] / buf[32 + i]))
