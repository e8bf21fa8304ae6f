"""Dev tool (GPU box): how the bench's re-draw loop converges, per precision."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
N = 1 << 20
P = W.perturbed_columns(N, seed=0); sc = torch.tensor(W.forcing_scale(N, seed=1000), device="cuda")
f = W.synth1_forcing()
engs = {}
for nm, dt in (("f32", torch.float32), ("f64", torch.float64)):
    engs[nm] = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300/3600., ponded_depth_max=0.0, dtype=dt)
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * sc[None, :]).contiguous()
pe = torch.zeros_like(pr)
for it in range(1, 12):
    st = {}
    for nm, e in engs.items():
        e.reset()
        e.forward(pr.to(e.dtype), pe.to(e.dtype), series=(), check=False)
        st[nm] = e.status.clone()
    bad = torch.nonzero((st["f32"] != 0) | (st["f64"] != 0)).flatten()
    print(it, "f32 faults", int((st["f32"] != 0).sum()), "f64 faults", int((st["f64"] != 0).sum()), "union", bad.numel(),
          "f32 bits", torch.unique(st["f32"]).tolist()[:8], "f64 bits", torch.unique(st["f64"]).tolist()[:8], flush=True)
    if bad.numel() == 0: break
    Q = W.perturbed_columns(int(bad.numel()), seed=100003 + it)
    for e in engs.values():
        for k, t in (("alpha", e.alpha), ("n", e.n), ("ksat", e.ksat), ("theta_e", e.theta_e), ("theta_r", e.theta_r)):
            t[:, bad] = torch.tensor(Q[k], device="cuda").to(t.dtype)
prev = None
for it in range(12, 16):
    st = {}
    for nm, e in engs.items():
        e.reset()
        e.forward(pr.to(e.dtype), pe.to(e.dtype), series=(), check=False)
        st[nm] = e.status.clone()
    for nm in ("f32", "f64"):
        b = torch.nonzero(st[nm] != 0).flatten().cpu().numpy()
        print(it, nm, len(b), "first", b[:6], "lane hist", np.bincount(b % 64, minlength=64)[:8], "n_fronts of bad", np.bincount(engs[nm].n_fronts.cpu().numpy()[b])[:12], flush=True)
        if nm == "f64":
            if prev is not None: print("   overlap with previous f64 bad set:", len(np.intersect1d(prev, b)))
            prev = b
    # no redraw: the same columns again -> deterministic?
