"""Do host <-> device copies overlap the forward kernels?  (dev tool, GPU box)  The forward kernels are PERSISTENT: their grid is
one wave per wave slot of the chip, so a copy that runs as a shader (a blit kernel) can only start when those waves end, while a
copy on the DMA engines overlaps.  Times 12 chunks of [48, 262144] fp32: kernels alone, copies alone (each way), both at once.
usage: python tools/overlap_probe.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W

N, rows, chunks = 262144, 48, 12
dev = torch.device("cuda:0")
P = W.perturbed_columns(N, seed=7)
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                    ponded_depth_max=0.0, dtype=torch.float32)
f = W.synth1_forcing(4)[: rows * chunks]
pr = torch.tensor(f[:, 0:1].astype(np.float32) * W.forcing_scale(N, 0.5, 1.0, seed=8).astype(np.float32)[None, :], device=dev)
pe = torch.zeros(rows, N, dtype=torch.float32, device=dev)
host = torch.empty(rows, N, dtype=torch.float32).pin_memory()
dst = [torch.empty(rows, N, dtype=torch.float32, device=dev) for _ in range(2)]
hback = torch.empty(rows, N, dtype=torch.float32).pin_memory()
side, back = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def kernels():
    for c in range(chunks):
        eng.forward(pr[c * rows:(c + 1) * rows], pe, series=(), basin=("runoff",), check=False)


def h2d():
    with torch.cuda.stream(side):
        for c in range(chunks):
            dst[c % 2].copy_(host, non_blocking=True)


def d2h():
    with torch.cuda.stream(back):
        for c in range(chunks):
            hback.copy_(dst[c % 2], non_blocking=True)


def timed(*fns):
    best = 1e9
    for _ in range(3):
        eng.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for fn in fns:
            fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 2)


out = {"env": {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES") if os.environ.get(k)},
       "kernels_ms": timed(kernels), "h2d_ms": timed(h2d), "d2h_ms": timed(d2h),
       "h2d_then_kernels_enqueued_ms": timed(h2d, kernels), "kernels_then_h2d_enqueued_ms": timed(kernels, h2d),
       "kernels_h2d_d2h_ms": timed(kernels, h2d, d2h)}
print(json.dumps(out))
