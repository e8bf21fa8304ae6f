"""First-contact GPU probe: parity of the HIP path vs golden vectors + rough timing.  (dev tool)"""
import glob, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lgar_py_amd as lg
from lgar_py_amd import ACC_NAMES

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
def run_case(name, dtype, ncol=3, mode=0):
    g = np.load(os.path.join(G, name + ".npz"))
    eng = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=ncol,
                        dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]), ponded_depth_max=float(g["pdm"]),
                        dtype=dtype, search_mode=mode)
    f = torch.tensor(g["forcing"])
    T = f.shape[0]
    pr = f[:, 0:1].expand(T, ncol).contiguous(); pe = f[:, 1:2].expand(T, ncol).contiguous()
    t0 = time.time()
    try:
        out = eng.forward(pr, pe, series=ACC_NAMES)
    except Exception as e:
        print(name, "EXC", e); return
    torch.cuda.synchronize(); dt = time.time() - t0
    acc = np.stack([out[nm][:, 0].cpu().numpy() for nm in ACC_NAMES], 1)
    d = np.abs(acc - g["acc"]); scale = np.maximum(np.abs(g["acc"]), 1e-6)
    rel = (d / scale).max(0)
    cum = np.abs(acc[:, :8].sum(0) - g["acc"][:, :8].sum(0)) / np.maximum(np.abs(g["acc"][:, :8].sum(0)), 1e-6)
    fr = eng.fronts()
    nfm = int(fr["n_fronts"][0] != g["nfronts"][-1])
    same = all(bool((out[nm][:, 0] == out[nm][:, ncol - 1]).all()) for nm in ACC_NAMES)
    print(f"{name:28s} mode{mode} {str(dtype)[6:]:8s} T={T} {dt:.2f}s maxrel/step {rel.max():.2e} cumrel {cum.max():.2e} nf_final_mismatch {nfm} replicas_equal {same}")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(G, "*.npz")) if "leaf" not in f)
    for nm in names:
        run_case(nm, torch.float64, mode=1)
    for nm in names:
        if nm.startswith("synth1") or nm.startswith("phil_pert"):
            run_case(nm, torch.float32, mode=1)
    # rough throughput: synth_1 tiled, perturbed params
    for dtype, N, mode in ((torch.float32, 1 << 18, 0), (torch.float32, 1 << 18, 1), (torch.float64, 1 << 16, 0), (torch.float64, 1 << 16, 1)):
        g = np.load(os.path.join(G, "synth1_phil.npz"))
        rng = np.random.default_rng(0)
        P = {k: torch.tensor(g[k][:, None] * (1 + 0.1 * (2 * rng.random((3, N)) - 1))) for k in ["alpha", "n", "ksat", "theta_e", "theta_r"]}
        th = torch.tensor(g["thickness"][:, None].repeat(N, 1))
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], th, dt_h=float(g["dt_h"]), ponded_depth_max=0.0, dtype=dtype, search_mode=mode)
        f = torch.tensor(g["forcing"]); T = f.shape[0]
        pr = f[:, 0:1].expand(T, N).contiguous().cuda().to(dtype); pe = f[:, 1:2].expand(T, N).contiguous().cuda().to(dtype)
        for rep in range(2):
            eng.reset(); torch.cuda.synchronize(); t0 = time.time()
            try:
                eng.forward(pr, pe, check=False)
            except Exception as e:
                print("EXC", e)
            torch.cuda.synchronize(); dt = time.time() - t0
            st = eng.status.cpu().numpy()
            print(f"throughput mode{mode} {str(dtype)[6:]} N={N} T={T}: {dt*1e3:.1f} ms -> {N*T/dt:.3e} col-steps/s; faulted {int((st!=0).sum())} bits {np.bitwise_or.reduce(st)}")
