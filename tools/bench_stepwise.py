"""Drop-in latency (dev tool): the reference's row-by-row agent loop, N = 1, Phillipsburg hourly, no_grad and grad mode."""
import os, sys, time, json, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_io import write_forcing, write_soil_dat
from lgar_py_amd import config
from lgar_py_amd.data import Data
from lgar_py_amd.model import MassBalance, dpLGAR
g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "data"))
n = 1500
cfg = config.load_config(cwd=tmp, overrides={"data.forcing_file": write_forcing(os.path.join(tmp, "data", "f.csv"), g["forcing"][:n]),
                                             "data.soil_params_file": write_soil_dat(os.path.join(tmp, "data", "s.dat")), "models.endtime": float(n)})
data = Data(cfg)
res = {}
# (round 2 timed "no_grad" first and cold: its 1.7 k steps/s against 3.0 k for "grad" was the one-time cost of the first launches
# -- code-object load, allocator warm-up -- spread over 300 steps, not a property of the mode.  Every mode now runs warm.)
for mode in ("warmup", "no_grad", "grad", "no_grad_again"):
    model = dpLGAR(cfg); mb = MassBalance(cfg, model)
    ctx = torch.enable_grad() if mode == "grad" else torch.no_grad()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ctx:
        ys = []
        for i in range(n):
            r, _ = model(data[i][0]); ys.append(r); mb.change_mass(model)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if mode == "grad":
        loss = torch.stack(ys).sum() + mb.AET * 0
        loss.backward(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    if mode != "warmup":
        res[mode] = dict(steps_per_s=n / (t1 - t0), us_per_step=1e6 * (t1 - t0) / n, backward_s=t2 - t1)
print(json.dumps(res))
