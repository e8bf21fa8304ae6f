import os, sys, time, tempfile, cProfile, pstats
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_io import write_forcing, write_soil_dat
from lgar_py_amd import config
from lgar_py_amd.data import Data
from lgar_py_amd.model import MassBalance, dpLGAR
g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "data"))
n = 300
cfg = config.load_config(cwd=tmp, overrides={"data.forcing_file": write_forcing(os.path.join(tmp, "data", "f.csv"), g["forcing"][:n]),
                                             "data.soil_params_file": write_soil_dat(os.path.join(tmp, "data", "s.dat")), "models.endtime": float(n)})
data = Data(cfg)
model = dpLGAR(cfg); mb = MassBalance(cfg, model)
mode = sys.argv[1] if len(sys.argv) > 1 else "no_grad"
with (torch.enable_grad() if mode == "grad" else torch.no_grad()):
    for i in range(20): model(data[i][0]); mb.change_mass(model)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); t0=time.perf_counter()
    for i in range(n): model(data[i][0]); mb.change_mass(model)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0; pr.disable()
print("steps/s", n/dt, "us/step", 1e6*dt/n)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
