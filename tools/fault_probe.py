import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
N = 1 << 17
P = W.perturbed_columns(N, seed=0)
f = W.synth1_forcing(); T = f.shape[0]
for dtype in (torch.float32,):
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300/3600, ponded_depth_max=0.0, dtype=dtype)
    pr = torch.tensor(f[:, 0:1]).expand(T, N).contiguous().cuda().to(dtype); pe = torch.zeros_like(pr)
    torch.cuda.synchronize(); t0 = time.time()
    eng.forward(pr, pe, check=False); torch.cuda.synchronize()
    st = eng.status.cpu().numpy()
    print(os.environ.get("LGAR_LIB"), dtype, "ms %.1f" % ((time.time()-t0)*1e3), "faulted", (st != 0).mean(), {b: int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 32, 64)})
    bad = np.nonzero(st)[0][:3]
    print("first bad", bad, st[bad])
