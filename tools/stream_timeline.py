"""Timeline of run_streamed_columns' pipeline (registered map, fixed chunk): GPU events around every copy and every forward,
CPU time per chunk."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
from lgar_py_amd.pipeline import close_forcing_file, open_forcing_file, write_forcing_file
N = 1 << 20
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = W.synth1_forcing(1); T = f.shape[0]
P = W.perturbed_columns(N, seed=7); sc = W.forcing_scale(N, 0.5, 1.0, seed=8).astype(np.float32)
d = tempfile.mkdtemp(prefix="lgar_tl_", dir="/dev/shm")
path = os.path.join(d, "precip.npy")
try:
    write_forcing_file(path, f[:, 0:1].astype(np.float32) * sc[None, :])
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float32)
    rm = open_forcing_file(path, register=True)
    dev = eng.device
    main = torch.cuda.current_stream(dev); side = torch.cuda.Stream(dev)
    bounds = [(lo, min(lo + CH, T)) for lo in range(0, T, CH)]
    bufs = [(torch.empty(CH, N, dtype=torch.float32, device=dev), torch.zeros(CH, N, dtype=torch.float32, device=dev)) for _ in range(NB)]
    import warnings
    warnings.simplefilter("ignore")
    for rep in range(3):
        eng.reset()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        c0 = [ev() for _ in bounds]; c1 = [ev() for _ in bounds]; k0 = [ev() for _ in bounds]; k1 = [ev() for _ in bounds]
        ready = [torch.cuda.Event() for _ in range(NB)]; freed = [torch.cuda.Event() for _ in range(NB)]
        for b in range(NB): freed[b].record(main)
        cpu = []
        torch.cuda.synchronize()
        t0 = time.perf_counter(); z = ev(); z.record(main)
        def upload(ci):
            lo, hi = bounds[ci]; b = ci % NB
            host = torch.from_numpy(rm[lo:hi])
            with torch.cuda.stream(side):
                side.wait_event(freed[b])
                c0[ci].record(side)
                bufs[b][0][:hi - lo].copy_(host, non_blocking=True)
                c1[ci].record(side)
                ready[b].record(side)
        for j in range(min(NB - 1, len(bounds))): upload(j)
        for ci, (lo, hi) in enumerate(bounds):
            b = ci % NB; n = hi - lo
            ta = time.perf_counter()
            main.wait_event(ready[b])
            k0[ci].record(main)
            out = eng.forward(bufs[b][0][:n], bufs[b][1][:n], series=(), basin=("runoff",), check=False)
            k1[ci].record(main)
            freed[b].record(main)
            tb = time.perf_counter()
            if ci + NB - 1 < len(bounds): upload(ci + NB - 1)
            tc = time.perf_counter()
            cpu.append((round(1e3 * (ta - t0), 3), round(1e3 * (tb - ta), 3), round(1e3 * (tc - tb), 3)))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if rep == 2:
            print(json.dumps(dict(chunk=CH, buffers=NB, wall_ms=round(1e3 * wall, 2))))
            for ci in range(len(bounds)):
                print(json.dumps(dict(ci=ci, copy=[round(z.elapsed_time(c0[ci]), 3), round(z.elapsed_time(c1[ci]), 3)],
                                      kern=[round(z.elapsed_time(k0[ci]), 3), round(z.elapsed_time(k1[ci]), 3)], cpu_start_fwd_upload=cpu[ci])))
    close_forcing_file(rm)
finally:
    if os.path.exists(path): os.remove(path)
    os.rmdir(d)
