"""Cycle attribution of the bench ensemble's forward launch (one lane per column, every SIMD full): the `clocks` measurement
variant (tools/ablate.py build with LGAR_VARIANTS=clocks) adds the shader-clock cycles between consecutive measurement points
of the grid's FIRST wave to one slot per point (csrc/lgar_measure.hpp LGAR_POINT_CLK); that wave shares its SIMD with the
kernel's other resident waves, so a slot holds wall cycles of the wave, the others' turns included.  Every point costs
~600 cycles itself.  usage: python tools/bigjob_clocks.py [mix|f64|f32] [columns]   (dev tool)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LGAR_LIB", os.path.join(ROOT, "lgar_py_amd", "csrc", "variants", "liblgar_hip_clocks.so"))
import torch
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

NAMES = {0: "step prologue (waits for the forcing load)", 9: "loop top: argument block, forcing load issued", 18: "accumulators -> series stores",
         21: "basin sums, per-call sums in LDS, drain", 1: "free-drainage front, AET", 2: "insert_water",
         3: "sweep: rest (check_column_mass, loop exit)", 4: "event scan / passes", 5: "create front, ponded depth",
         6: "calc_dzdt: loop exit", 7: "mass_balance", 8: "GIUH, NaN check",
         10: "calc_dzdt: find the next moving front", 11: "calc_dzdt: pick, Se (mixed) / fused ends begin", 12: "trapezoid: K of both ends",
         13: "trapezoid: set-up, wave-uniform loop bounds", 14: "trapezoid: interior nodes", 16: "trapezoid: tail group, closing",
         17: "calc_dzdt: conductivities above, dz/dt arithmetic", 19: "sweep: loop control", 20: "sweep: layer-bottom front (psi continuity)",
         22: "sweep: in-layer front, thetas before the search", 23: "sweep: search + theta", 24: "sweep: psi from theta, carry",
         25: "fused trapezoid: begin", 26: "fused trapezoid: ends", 29: "fused trapezoid: nodes"}
which = sys.argv[1] if len(sys.argv) > 1 else "mix"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dtype = torch.float32 if which == "f32" else torch.float64
kw = {"geff_precision": "f32"} if which == "mix" else {}
f = W.synth1_forcing()
T = f.shape[0]
P = W.perturbed_columns(N, seed=0)
sc = torch.tensor(W.forcing_scale(N, seed=1000), device="cuda")
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                    ponded_depth_max=0.0, dtype=dtype, **kw)
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * sc[None, :]).to(dtype).contiguous()
pe = torch.zeros_like(pr)
out = {"runoff": torch.empty(T, N, dtype=dtype, device="cuda"), "percolation": torch.empty(T, N, dtype=dtype, device="cuda")}
buf = (ctypes.c_ulonglong * 64)()
for rep in range(2):
    eng.reset()
    eng.lib.lgar_debug_clocks(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.forward(pr, pe, series=("runoff", "percolation"), out=out, check=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
assert eng.lib.lgar_debug_clocks(buf, 0) == 0
tot = sum(buf[i] for i in range(32))
steps = buf[32 + 8] or 1  # wave-steps of the first wave (point 8 is hit once per sub-step)
print(json.dumps(dict(job=which, columns=N, ms=round(ms, 2), cycles_total=tot, wave_steps=steps, cycles_per_step=round(tot / steps, 1))))
for i in range(32):
    if buf[32 + i]:
        print("%2d %-62s %6.2f %%  %9.1f cycles/step  %8.2f hits/step  %8.1f cycles/hit" % (
            i, NAMES.get(i, "?"), 100.0 * buf[i] / tot, buf[i] / steps, buf[32 + i] / steps, buf[i] / buf[32 + i]))
