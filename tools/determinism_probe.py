"""Dev tool (GPU box): the same 1M-column job three times -- identical status words and series?  usage: determinism_probe.py [f32|f64|mix]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "f32"
N = 1 << 20
P = W.perturbed_columns(N, seed=0); sc = torch.tensor(W.forcing_scale(N, seed=1000), device="cuda")
f = W.synth1_forcing()
dt = torch.float32 if which == "f32" else torch.float64
kw = dict(geff_precision="f32") if which == "mix" else {}
e = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300/3600., ponded_depth_max=0.0, dtype=dt, **kw)
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * sc[None, :]).to(dt).contiguous(); pe = torch.zeros_like(pr)
import hashlib
ref = None
for it in range(3):
    e.reset()
    o = e.forward(pr, pe, series=("runoff",), check=False)
    cur = (e.status.clone(), o["runoff"].clone(), e.theta.clone())
    if ref is None:
        ref = cur
        h = hashlib.sha256()
        for t in cur:
            h.update(t.cpu().numpy().tobytes())
        print(which, "digest of run 0 (status words, runoff series, front-table theta):", h.hexdigest()[:16], flush=True)
    else:
        print(which, "run", it, "status differs in", int((cur[0] != ref[0]).sum()), "columns; runoff differs in", int((cur[1] != ref[1]).any(0).sum()),
              "theta differs in", int(((cur[2] != ref[2]) & ~(torch.isnan(cur[2]) & torch.isnan(ref[2]))).any(0).sum()), "faults", int((cur[0] != 0).sum()), flush=True)
