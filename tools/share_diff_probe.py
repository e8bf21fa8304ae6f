import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
for N in (512, 500):
  for width in (8, 9):
    E = W.ensemble_columns(N, seed=3)
    f = W.synth1_forcing(); T = f.shape[0]
    pr = torch.tensor(f[:, 0:1], device="cuda"); pe = torch.zeros_like(pr)
    torch.manual_seed(0)
    w = torch.rand(T, 1, device="cuda", dtype=torch.float64)
    rep = {k: np.repeat(v, width, axis=1) for k, v in E.items()}
    eng = lg.LgarEngine(rep["alpha"], rep["n"], rep["ksat"], rep["theta_e"], rep["theta_r"], rep["thickness"],
                        dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float64, with_state=False)
    dirs = {k: torch.zeros(3, width * N, dtype=torch.float64, device="cuda") for k in ("alpha", "n", "ksat")}
    basis = [(kind, l) for kind in ("alpha", "n", "ksat") for l in range(3)]
    for b in range(width):
        kind, l = basis[b % 9]; dirs[kind][l, b::width] = 1.0
    g0, _, s0 = eng.tangent(dirs, pr, pe, w_runoff=w, forcing_group=width)
    g1, _, s1 = eng.tangent(dirs, pr, pe, w_runoff=w, forcing_group=width, share=width)
    ok = (s0 & 0x7F) == 0
    d = (g0 - g1).abs() * ok
    i = int(d.argmax())
    print("N", N, "W", width, "status equal", bool(torch.equal(s0, s1)), "max diff %.3e at big-col %d (col %d dir %d) g0 %.6e g1 %.6e  scale %.3e; n bad>1e-9*scale: %d" % (
        float(d.max()), i, i // width, i % width, float(g0[i]), float(g1[i]), float(g0[ok].abs().max()), int((d > 1e-9 * float(g0[ok].abs().max())).sum())))
    rel = (d / g0.abs().clamp_min(1e-12))[ok]
    print("   max diff relative to own magnitude %.3e, median %.3e" % (float(rel.max()), float(rel.median())))
