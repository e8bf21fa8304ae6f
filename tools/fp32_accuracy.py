"""fp32 (throughput configuration) vs fp64 on the same seeded bench columns: error distribution of run totals (dev tool)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
P = W.perturbed_columns(N, seed=0); sc = W.forcing_scale(N, seed=1000); f = W.synth1_forcing(); T = f.shape[0]
res = {}
for dt in (torch.float64, torch.float32):
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300 / 3600, ponded_depth_max=0.0, dtype=dt)
    pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[None, :]).to(dt).contiguous()
    out = eng.forward(pr, torch.zeros_like(pr), series=("runoff",), basin=("runoff",), check=False)
    res[dt] = (eng.totals.double().cpu().numpy(), eng.status.cpu().numpy(), out["basin:runoff"].cpu().numpy())
t64, s64, b64 = res[torch.float64]; t32, s32, b32 = res[torch.float32]
ok = (s64 == 0) & (s32 == 0)
rep = dict(columns=N, valid_in_both=float(ok.mean()), flagged_fp64=float((s64 != 0).mean()), flagged_fp32=float((s32 != 0).mean()))
for j, nm in ((3, "infiltration"), (4, "runoff"), (9, "ending_volume")):
    scale = np.maximum(np.abs(t64[0][ok]), 1.0) if nm == "runoff" else np.maximum(np.abs(t64[j][ok]), 1e-6)
    r = np.abs(t32[j][ok] - t64[j][ok]) / scale
    rep[nm] = dict(median=float(np.median(r)), p90=float(np.percentile(r, 90)), p99=float(np.percentile(r, 99)), max=float(r.max()))
rep["basin_runoff_rel_diff"] = float(abs(b32.sum() - b64.sum()) / abs(b64.sum()))
print(json.dumps(rep))
