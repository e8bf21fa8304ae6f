#!/usr/bin/env python3
"""bench.py -- column-timesteps/s of the LGAR hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2]/[3]): per GPU, 1M synthetic columns
(Phillipsburg P-1..3 soils perturbed +-10 % per column, forcing_data_synth_1 shape x per-column
scale U(0.5,1.5)), 3 layers, fp32, T = 144 five-minute steps per pass.  One "step" = one pass of
the hot path over that batch: set_internal_states + 144 x forward() for every column (one kernel
launch with the time loop inside) + the basin-runoff reduction [T] (RCCL all-reduce when N > 1).
Inputs are resident in HBM before the timed region.  Weak scaling: columns per GPU fixed.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def alg_bytes_per_col_step(elem, T, L=3, fmax=16):
    """B_alg = 4*s + B_col/T; B_col = L*6*s + 2*F_MAX*(5*s+1) + 2*16*s."""
    b_col = L * 6 * elem + 2 * fmax * (5 * elem + 1) + 2 * 16 * elem
    return 4 * elem + b_col / T


def measured_traffic(N, T, dtype):
    """HBM bytes per launch from the committed PMC passes (profiles/*/traffic.json), when they were taken on exactly
    this workload; FETCH_SIZE corrected x2 as MI355X_MICROARCH.md prescribes for gfx950.  None otherwise."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, r, "traffic.json")
        if os.path.exists(f):
            t = json.load(open(f))
            if (t.get("columns"), t.get("timesteps"), t.get("dtype")) == (N, T, dtype):
                best = (2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024
    return best


def measured_valu(N, T, dtype):
    """Compute-side roof from the committed PMC passes (profiles/*/traffic.json 'valu' block), same workload only."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, r, "traffic.json")
        if os.path.exists(f):
            t = json.load(open(f))
            if (t.get("columns"), t.get("timesteps"), t.get("dtype")) == (N, T, dtype) and "valu" in t:
                best = t["valu"]
    return best


def cpu_baseline(target_s=12.0):
    """Oracle (C restatement, fp64, OpenMP over columns) timed on this host on a bounded sample of the
    same workload.  A reported baseline, never the thing measured above."""
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O

    # the GPU box gives one GPU a 16-core CPU share (more OpenMP threads than that only oversubscribe the cgroup)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("LGAR_CPU_THREADS", min(avail, 16)))
    f = W.synth1_forcing()
    T = f.shape[0]

    def run(n):
        P = W.perturbed_columns(n, seed=0)
        sc = W.forcing_scale(n, seed=1)
        pr = f[:, 0:1] * sc[None, :]
        pe = np.zeros_like(pr)
        t0 = time.perf_counter()
        O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                      pdm=0.0, dt_h=300.0 / 3600.0, nthreads=cores, want_series=False)
        return time.perf_counter() - t0

    n0 = 64 * cores
    t = run(n0)
    n = int(max(n0, min(n0 * target_s / max(t, 1e-3), 2_000_000)))
    n = (n // 64) * 64
    t = run(n)
    return {"value": n * T / t, "unit": "column-timesteps/s", "cores": cores, "kind": "port",
            "sample": "%d columns x %d steps of the same synth_1 workload, fp64 C oracle, OpenMP, %.1f s" % (n, T, t),
            "reference_pytorch_cpu_loop": "10.9 column-timesteps/s on this forcing shape, 1 core (BASELINE.md section 2: observed "
                                          "in the survey container; the Python reference cannot travel to the GPU box)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--columns", type=int, default=1 << 20, help="columns per GPU")
    ap.add_argument("--tile", type=int, default=1, help="time tiling of the 144-row synth_1 forcing")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="synth1", choices=["synth1", "phillipsburg"],
                    help="synth1 (default): BASELINE configs[2]/[3]; phillipsburg: configs[1], replicated Phillipsburg "
                         "column x 3000 hourly steps (use with --columns 10000 --dtype f64)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the LGAR engine has no CPU fallback")
    import torch.distributed as dist

    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W

    # one process per GPU; LGAR_DIST_BACKEND=gloo is a rehearsal mode (several ranks may then share one GPU)
    backend = os.environ.get("LGAR_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:  # gloo rehearsal: reduce on the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    elem = 4 if args.dtype == "f32" else 8
    N = args.columns
    if args.workload == "phillipsburg":
        g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))  # first 3000 rows of the bundled forcing
        f = g["forcing"]
        T = f.shape[0]
        P = {k: np.repeat(np.asarray(W.PHILLIPSBURG[k], dtype=np.float64)[:, None], N, 1) for k in
             ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")}
        sc = torch.ones(N, dtype=torch.float64, device=dev)
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                            dt_h=1.0, ponded_depth_max=2.0, dtype=dtype, device=dev)
    else:
        f = W.synth1_forcing(args.tile)
        T = f.shape[0]
        P = W.perturbed_columns(N, seed=rank)  # rank r holds columns [r*N, (r+1)*N) of the job; seed = shard index
        sc = torch.tensor(W.forcing_scale(N, seed=1000 + rank), device=dev)
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                            dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=dtype, device=dev)
    precip = (torch.tensor(f[:, 0], device=dev)[:, None] * sc[None, :]).to(dtype).contiguous()
    pet = (torch.tensor(f[:, 1], device=dev)[:, None] * torch.ones_like(sc)[None, :]).to(dtype).contiguous()
    # Keep the ensemble inside the reference's domain of validity (untimed set-up): a perturbed column whose
    # run makes the reference raise (status != 0, e.g. the negative-pow-base path of insert_water)
    # gets its soil re-drawn from the same distribution until no column faults.
    resampled = 0
    for it in range(1, 16):
        eng.reset()
        eng.forward(precip, pet, series=(), check=False)
        bad = torch.nonzero(eng.status != 0).flatten()
        if bad.numel() == 0:
            break
        resampled += int(bad.numel())
        Q = W.perturbed_columns(int(bad.numel()), seed=(rank + 1) * 100003 + it)
        for k, t in (("alpha", eng.alpha), ("n", eng.n), ("ksat", eng.ksat), ("theta_e", eng.theta_e), ("theta_r", eng.theta_r)):
            t[:, bad] = torch.tensor(Q[k], device=dev).to(dtype)
    out = {"runoff": torch.empty(T, N, dtype=dtype, device=dev), "percolation": torch.empty(T, N, dtype=dtype, device=dev)}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        eng.reset()
        if i is not None:
            ev[i][0].record()  # same stream as the kernel launch (torch's current stream)
        res = eng.forward(precip, pet, series=("runoff", "percolation"), out=out, basin=("runoff",), check=False)
        if i is not None:
            ev[i][1].record()
        basin = res["basin:runoff"]  # basin runoff per timestep [T] (fp64), reduced in the kernel epilogue
        if world > 1:
            all_reduce(basin)  # the only exchange of the path
        return basin

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    basin = None
    for i in range(args.steps):
        basin = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    faulted = torch.tensor([int((eng.status != 0).sum().item())], device=dev)
    if world > 1:
        all_reduce(faulted)

    if rank == 0:
        units = N * world * T * args.steps
        b_alg = alg_bytes_per_col_step(elem, T)
        achieved = b_alg * N * T / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "column-timesteps/sec", "value": units / elapsed, "unit": "column-timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]/[3]: %d synthetic columns per GPU (Phillipsburg P-1..3 soils "
                                    "+-10%% per column, forcing_data_synth_1 shape x U(0.5,1.5) per column), 3 layers, %s, "
                                    "T=%d steps of 300 s per pass; pass = set_internal_states + T x forward + basin-runoff "
                                    "reduction" % (N, args.dtype, T)) if args.workload == "synth1" else
                                   ("BASELINE configs[1]: %d replicated Phillipsburg columns, 3 layers, %s, T=%d hourly steps "
                                    "(bundled forcing, pdm 2 cm) per pass" % (N, args.dtype, T)),
                       "columns_per_gpu": N, "timesteps_per_pass": T, "columns_redrawn_to_stay_in_reference_domain": resampled, "parallelism": "columns sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(N, T, args.dtype),
                         "traffic_note": "bytes per launch from separate rocprofv3 --pmc passes (profiles/r01/traffic.json), not live",
                         "algorithmic_bytes_per_launch": b_alg * N * T,
                         "kernel": "lgar_forward_kernel<%s,3,12>" % ("float" if elem == 4 else "double"),
                         "kernel_ms": kern_ms, "alg_bytes_per_column_timestep": b_alg,
                         "note": "path is VALU bound, not HBM bound (~1e3 flop/B; see valu_roofline and DESIGN.md); "
                                 "alg bytes use SURVEY 8(d)'s figure (F_MAX=16 state); the kernel's own state is F_MAX=12"},
            "valu_roofline": measured_valu(N, T, args.dtype),
            "faulted_columns": int(faulted.item()),
            "basin_runoff_total_cm": float(basin.sum().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
