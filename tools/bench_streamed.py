"""Long-run demo (dev tool): 1M perturbed columns x 3000 hourly Phillipsburg steps (rain + PET), streamed in chunks."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
from lgar_py_amd.pipeline import run_streamed
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "phil_hourly_3000.npz"))
x = g["forcing"]; T = x.shape[0]
P = W.perturbed_columns(N, seed=0); sc = W.forcing_scale(N, 0.5, 1.5, seed=1)
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=1.0, ponded_depth_max=2.0, dtype=dtype)
for rep in range(2):
    eng.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        basin = run_streamed(eng, x, scale=sc, chunk=250, series=("runoff", "AET"))
    except ValueError as e:
        basin = None; msg = str(e)[:80]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = eng.status.cpu().numpy()
print(json.dumps(dict(columns=N, T=T, dtype=str(dtype), seconds=dt, col_steps_per_s=N * T / dt, faulted_fraction=float((st != 0).mean()),
                      aet_total_mean_cm=float(eng.totals[2].double().mean()), max_fronts=int(eng.n_fronts.max()))))
print(json.dumps({"status_bit_counts": {str(b): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 32, 64)}, "overflow_columns": int(((st & 8) != 0).sum()),
                  "n_fronts_hist": np.bincount(eng.n_fronts.cpu().numpy(), minlength=33).tolist()}))
