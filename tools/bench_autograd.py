"""BASELINE configs[4]: 100k-column vG parameter ensemble, forward + backward through the HIP kernels (dev tool).
Prints fwd and fwd+bwd column-timesteps/s."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lgar_py_amd import workloads as W
from lgar_py_amd.autograd import lgar_series

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dtype = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
E = W.ensemble_columns(N, seed=0)
f = W.synth1_forcing(); T = f.shape[0]
pr = torch.tensor(f[:, 0:1], device="cuda").expand(T, N).contiguous().to(dtype); pe = torch.zeros_like(pr)
P = {k: torch.tensor(v, device="cuda", dtype=dtype) for k, v in E.items()}
for k in ("alpha", "n", "ksat"):
    P[k].requires_grad_(True)
res = {}
for rep in range(2):
    for k in ("alpha", "n", "ksat"):
        P[k].grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = []
    runoff, _ = lgar_series(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                            dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=dtype, check=False, status_out=st)
    ok = st[0] == 0
    loss = torch.mean(runoff[:, ok] ** 2)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = dict(columns=N, T=T, dtype=str(dtype), valid_fraction=float(ok.double().mean()), loss=float(loss.detach()),
               tangent_faulted_fraction=float((st[1] != 0).double().mean()),
               fwd_ms=1e3 * (t1 - t0), bwd_ms=1e3 * (t2 - t1), fwd_col_steps_per_s=N * T / (t1 - t0),
               fwd_bwd_col_steps_per_s=N * T / (t2 - t0))
print(json.dumps(res))
