"""Per-column soil-water closure of the bench ensemble (dev tool): start volume + infiltration - AET - percolation - end
volume over the run, fp32 and fp64, quantiles and the worst columns."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
P = W.perturbed_columns(N, seed=0)
sc = W.forcing_scale(N, seed=1000)
f = W.synth1_forcing()
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                        ponded_depth_max=0.0, dtype=dt)
    pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[None, :]).to(dt).contiguous()
    start = eng.ending_volume.double().clone()
    eng.forward(pr, torch.zeros_like(pr), series=(), check=False)
    tot = eng.totals.double()
    end = eng.ending_volume.double()
    soil = (start + tot[3] - tot[2] - tot[5] - end).abs()
    ok = eng.status == 0
    s = soil[ok].cpu().numpy()
    idx = torch.nonzero(ok).flatten().cpu().numpy()
    worst = np.argsort(-s)[:8]
    res[name] = dict(valid=float(ok.double().mean()), q50=float(np.quantile(s, .5)), q99=float(np.quantile(s, .99)),
                     q9999=float(np.quantile(s, .9999)), max=float(s.max()), n_above_1e_3=int((s > 1e-3).sum()),
                     n_above_1e_4=int((s > 1e-4).sum()),
                     worst=[dict(col=int(idx[w]), err=float(s[w]), end=float(end[idx[w]]), infil=float(tot[3][idx[w]]),
                                 nf=int(eng.n_fronts[idx[w]])) for w in worst])
print(json.dumps(res, indent=1))
