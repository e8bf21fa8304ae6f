"""Where a small job's time goes: 64 replicated Phillipsburg columns x 3000 h (one wavefront, fp64) on the measurement variants
built by tools/ablate.py-style flags (csrc/variants/).  (dev tool)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
g = np.load(os.path.join(%(root)r, "tests", "golden", "phil_hourly_3000.npz")); f = g["forcing"]; P = W.PHILLIPSBURG
N, T = 64, 3000
for lanes in (1, 64):
    eng = lg.LgarEngine(*[P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=N, dt_h=1.0,
                        ponded_depth_max=2.0, dtype=torch.float64, forward_lanes=lanes)
    pr = torch.tensor(f[:T, 0:1], device="cuda").expand(T, N).contiguous(); pe = torch.tensor(f[:T, 1:2], device="cuda").expand(T, N).contiguous()
    best = 1e9
    for _ in range(2):
        eng.reset(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.forward(pr, pe, series=("runoff",), check=False); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(json.dumps(dict(variant=%(name)r, lanes=lanes, ms=round(best, 2), us_per_step=round(best * 1e3 / T, 2))), flush=True)
"""
vd = os.path.join(ROOT, "lgar_py_amd", "csrc", "variants")
names = ["base"] + sorted(f[len("liblgar_hip_"):-3] for f in os.listdir(vd) if f.endswith(".so"))
for nm in names:
    env = dict(os.environ)
    if nm != "base":
        env["LGAR_LIB"] = os.path.join(vd, "liblgar_hip_%s.so" % nm)
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, name=nm)], env=env, capture_output=True, text=True, timeout=600)
    print(p.stdout.strip() or p.stderr[-300:], flush=True)
