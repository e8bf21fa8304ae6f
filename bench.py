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

Without a launcher (`WORLD_SIZE` unset) and N > 1 the script starts its own N ranks, one fresh child process per GPU,
BEFORE anything in this process touches the GPU.  Prints ONE JSON line on rank 0.
LGAR_DIST_BACKEND=gloo is a rehearsal mode: the ranks may share GPUs and the [T] reduction goes through the host.
LGAR_FORCE_DIST=1 makes `--gpus 1` take the multi-rank code path too: a fresh child process joins an RCCL ("nccl") group of
world size 1 and every pass runs the real dist.all_reduce on the device [T] vector and the barriers of the timed region --
what a one-GPU box can execute of configs[3]'s collective (the JSON line then says so under config.collective).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def alg_bytes_per_col_step(elem, T, L=3, fmax=16):
    """B_alg = 4*s + B_col/T; B_col = L*6*s + 2*F_MAX*(5*s+1) + 2*16*s."""
    b_col = L * 6 * elem + 2 * fmax * (5 * elem + 1) + 2 * 16 * elem
    return 4 * elem + b_col / T


def library_fingerprint():
    """Content fingerprint of the HIP library's sources, headers and flags (lgar_py_amd/build.py): what a committed profile
    is checked against."""
    from lgar_py_amd import build as B
    return B._fingerprint()


def _matching_profile(N, T, dtype):
    """The latest committed PMC record (profiles/*/traffic*.json) taken on exactly this workload, with whether it was taken
    with THIS library build."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for r in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not os.path.isdir(os.path.join(pdir, r)):
            continue
        for fn in sorted(os.listdir(os.path.join(pdir, r))):
            if fn.startswith("traffic") and fn.endswith(".json"):
                with open(os.path.join(pdir, r, fn)) as fh:
                    t = json.load(fh)
                if (t.get("columns"), t.get("timesteps"), t.get("dtype")) == (N, T, dtype):
                    best = dict(t, source="profiles/%s/%s" % (r, fn))
    if best is not None:
        best["same_library"] = best.get("library_fingerprint") == library_fingerprint()
    return best


def measured_traffic(N, T, dtype):
    """(HBM bytes per launch, provenance) from the committed PMC passes, FETCH_SIZE corrected x2 as MI355X_MICROARCH.md
    prescribes for gfx950 -- or (None, reason): the counters come from separate rocprofv3 --pmc passes, not from this run, so
    the figure is only reported when the profile was taken on exactly this workload AND with this library build."""
    t = _matching_profile(N, T, dtype)
    if t is None:
        return None, "no committed profile of this workload (%d columns x %d steps, %s)" % (N, T, dtype)
    if not t["same_library"]:
        return None, "%s was taken with another build of the library (fingerprint %s, this build %s)" % (
            t["source"], str(t.get("library_fingerprint"))[:12], library_fingerprint()[:12])
    return (2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024, "%s (separate rocprofv3 --pmc passes on this library build)" % t["source"]


def measured_valu(N, T, dtype):
    """Compute-side counters from the committed PMC passes ('valu' block), same workload and same library build only."""
    t = _matching_profile(N, T, dtype)
    if t is None or "valu" not in t or not t["same_library"]:
        return None
    return dict(t["valu"], source=t["source"])


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(target_s=12.0):
    """Oracle (C restatement, fp64) timed on this host on a bounded sample of the same workload: all host cores of this
    box's CPU share (OpenMP over columns) and one thread.  A reported baseline, never the thing measured above."""
    import numpy as np

    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O

    # the GPU box gives one GPU a 16-core CPU share (more OpenMP threads than that only oversubscribe the cgroup)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("LGAR_CPU_THREADS", min(avail, 16)))
    f = W.synth1_forcing()
    T = f.shape[0]

    def run(n, threads):
        P = W.perturbed_columns(n, seed=0)
        sc = W.forcing_scale(n, seed=1)
        pr = f[:, 0:1] * sc[None, :]
        pe = np.zeros_like(pr)
        t0 = time.perf_counter()
        O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                      pdm=0.0, dt_h=300.0 / 3600.0, nthreads=threads, want_series=False)
        return time.perf_counter() - t0

    def sized(threads, budget):
        n0 = 64 * threads
        t = run(n0, threads)
        n = int(max(n0, min(n0 * budget / max(t, 1e-3), 2_000_000)))
        n = max(64, (n // 64) * 64)
        return n, run(n, threads)

    n, t = sized(cores, target_s * 0.7)
    n1, t1 = sized(1, target_s * 0.3)
    return {"value": n * T / t, "unit": "column-timesteps/s", "cores": cores, "kind": "port",
            "sample": "%d columns x %d steps of the same synth_1 workload, fp64 C oracle, OpenMP, %.1f s" % (n, T, t),
            "single_thread": {"value": n1 * T / t1, "cores": 1,
                              "sample": "%d columns x %d steps, 1 thread, %.1f s" % (n1, T, t1)},
            "cpu_model": cpu_model(), "host_cores_visible": avail,
            "torch_eager_standin": torch_eager_standin(min(3.0, target_s * 0.25)),
            "reference_pytorch_cpu_loop": "10.9 column-timesteps/s on this forcing shape, 1 core (BASELINE.md section 2: observed "
                                          "in the survey container; the Python reference cannot travel to the GPU box)"}


def torch_eager_standin(target_s=3.0):
    """stand-in for "the reference's PyTorch CPU loop" on THIS host.  The reference itself cannot
    travel to the GPU box; what can be re-established here is the cost of its per-column-timestep torch work: SURVEY.md
    section 3.4 counted, per column-timestep of the hourly Phillipsburg run, ~840 0-d fp64 torch.pow calls (safe_pow) plus
    their guards (973 isclose, 2043 isnan, 2043 any) -- 64 % + 25 % of its time sits in those leaf calls.  This replays
    exactly that op mix (plus 4000 scalar add/mul/sub/div, the remaining arithmetic) on 0-d tensors, one thread, and
    reports steps/s; the reference measured 12.9 (no_grad) on the survey container's core."""
    import torch
    torch.set_num_threads(1)
    a = torch.tensor(0.37, dtype=torch.float64)
    b = torch.tensor(1.61, dtype=torch.float64)
    z = torch.tensor(0.0, dtype=torch.float64)

    def one_step():
        x = a
        for _ in range(840):
            x = torch.pow(a, b)
        for _ in range(973):
            torch.isclose(x, z)
        for _ in range(2043):
            torch.isnan(x).any()
        for _ in range(1000):
            x = (a + b) * a - b / a
        return x

    with torch.no_grad():
        one_step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < target_s:
            one_step()
            n += 1
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "column-timesteps/s", "cores": 1, "kind": "op-mix replay (leaf torch ops only: an UPPER bound "
            "on the reference loop's throughput on this host; its Python-level bookkeeping is not replayed)",
            "sample": "%d replays of the reference's per-column-timestep torch op mix (840 pow, 973 isclose, 2043 isnan+any, "
                      "4000 scalar ops on 0-d fp64 tensors), %.1f s" % (n, dt)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--columns", type=int, default=1 << 20, help="columns per GPU")
    ap.add_argument("--tile", type=int, default=1, help="time tiling of the 144-row synth_1 forcing")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp64 / configs[1] sub-records and the VALU probe")
    ap.add_argument("--workload", default="synth1", choices=["synth1", "phillipsburg"],
                    help="synth1 (default): BASELINE configs[2]/[3]; phillipsburg: configs[1], replicated Phillipsburg "
                         "column x 3000 hourly steps (use with --columns 10000 --dtype f64)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """No launcher: start --gpus fresh child interpreters, one rank per GPU (this process has not touched the GPU and
    never will: it only waits).  Rank 0's stdout carries the JSON line."""
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, failed = wait_ranks(procs)
    sys.stdout.write(out0)
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were stopped\n" % failed)
        raise SystemExit(failed[1] if 0 < failed[1] < 256 else 1)


def wait_ranks(procs, poll_s=0.05):
    """Wait for the rank processes (rank 0's stdout is a pipe).  Returns (rank 0's output, None) when all exit with 0, else
    (output so far, (rank, code)) of the first rank seen to fail -- after stopping its siblings, which would otherwise sit in
    init_process_group / all_reduce until the RCCL timeout."""
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read() if procs[0].stdout else ""), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc is not None and rc != 0:
                failed = (r, rc)
                break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    if failed is None:
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
        failed = bad[0] if bad else None
    return "".join(c for c in chunks if c), failed


def valu_probe_record(dev):
    """Live compute-side roof on this chip: peak wave-instructions/s of v_exp_f32 / v_log_f32 / v_fma_f32 and of the
    Geff loop's own instruction mix (csrc/lgar_probe.hip), best over 2-4 resident waves per SIMD."""
    import ctypes as C

    import torch

    from lgar_py_amd import _capi
    lib = _capi.load()
    ops = {"v_exp_f32": 0, "v_log_f32": 1, "v_fma_f32": 4, "geff_mix": 16, "v_fma_f64": 12}
    rec = {}
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    for nm, op in ops.items():
        best = 0.0
        for w in (2, 4):
            lds = ((160 * 1024) // (4 * w) // 256) * 256
            nwg, iters = n_cu * 4 * w * 2, 1500
            sink = torch.zeros(nwg * 64, dtype=torch.float32, device=dev)
            for _ in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = lib.lgar_valu_probe(op, nwg, lds, iters, sink.data_ptr(), st)
                b.record()
                torch.cuda.synchronize(dev)
                if rc == 0:
                    best = max(best, nwg * iters * lib.lgar_valu_probe_insts(op) / (a.elapsed_time(b) * 1e-3))
        rec[nm] = best
    return rec


def main():
    args = parse_args()
    force_dist = os.environ.get("LGAR_FORCE_DIST", "0") == "1"
    if (args.gpus > 1 or force_dist) and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the LGAR engine has no CPU fallback")
    import torch.distributed as dist

    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W

    # one process per GPU; LGAR_DIST_BACKEND=gloo is a rehearsal mode (several ranks may then share one GPU)
    backend = os.environ.get("LGAR_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("--gpus %d but only %d visible (LGAR_DIST_BACKEND=gloo rehearses more ranks than GPUs)" % (world, ndev))
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist_on = world > 1 or force_dist  # LGAR_FORCE_DIST=1: the collective path with a group of one
    if dist_on:
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

    def make_workload(kind, n_cols, dt, seed_rank, **eng_kw):
        """(engine, precip[T, N], pet[T, N], redrawn) with inputs resident in HBM."""
        if kind == "phillipsburg":
            g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))  # first 3000 rows of the bundled forcing
            f = g["forcing"]
            P = {k: np.repeat(np.asarray(W.PHILLIPSBURG[k], dtype=np.float64)[:, None], n_cols, 1) for k in
                 ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")}
            sc = torch.ones(n_cols, dtype=torch.float64, device=dev)
            eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                                dt_h=1.0, ponded_depth_max=2.0, dtype=dt, device=dev, **eng_kw)
        else:
            f = W.synth1_forcing(args.tile)
            P = W.perturbed_columns(n_cols, seed=seed_rank)  # rank r holds columns [r*N, (r+1)*N) of the job; seed = shard index
            sc = torch.tensor(W.forcing_scale(n_cols, seed=1000 + seed_rank), device=dev)
            eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                                dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=dt, device=dev, **eng_kw)
        pr = (torch.tensor(f[:, 0], device=dev)[:, None] * sc[None, :]).to(dt).contiguous()
        pe = (torch.tensor(f[:, 1], device=dev)[:, None] * torch.ones_like(sc)[None, :]).to(dt).contiguous()
        # Keep the ensemble inside the reference's domain of validity (untimed set-up): a perturbed column whose
        # run makes the reference raise (status != 0, e.g. the negative-pow-base path of insert_water)
        # gets its soil re-drawn from the same distribution until no column faults -- in the timed precision AND in
        # fp64, the reference's own arithmetic (which of the two flags a column hinges on a psi tie at the 1e-8 level),
        # so that every timed column is one the reference integrates to the end.
        redrawn = 0
        chk = None
        if kind != "phillipsburg":
            other = torch.float64 if dt == torch.float32 else torch.float32
            chk = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                                dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=other, device=dev)
            pr2, pe2 = pr.to(other), pe.to(other)
        # the set-up launches do exactly what the timed ones do (same series and basin outputs), so that per-launch medians
        # of a profile of this command (rocprofv3 --stats / --pmc) are medians of the timed work
        scratch = {k: torch.empty(pr.shape[0], n_cols, dtype=dt, device=dev) for k in ("runoff", "percolation")}
        for it in range(1, 33):
            eng.reset()
            eng.forward(pr, pe, series=("runoff", "percolation"), out=scratch, basin=("runoff",), check=False)
            flagged = eng.status != 0
            if chk is not None:
                chk.reset()
                chk.forward(pr2, pe2, series=(), check=False)
                flagged |= chk.status != 0
            bad = torch.nonzero(flagged).flatten()
            if bad.numel() == 0 or kind == "phillipsburg":
                break
            redrawn += int(bad.numel())
            Q = W.perturbed_columns(int(bad.numel()), seed=(seed_rank + 1) * 100003 + it)
            for e in (eng, chk):
                for k, t in (("alpha", e.alpha), ("n", e.n), ("ksat", e.ksat), ("theta_e", e.theta_e), ("theta_r", e.theta_r)):
                    t[:, bad] = torch.tensor(Q[k], device=dev).to(t.dtype)
        else:
            raise RuntimeError("bench set-up: %d columns still outside the reference's domain after 32 re-draws" % bad.numel())
        del chk, scratch
        return eng, pr, pe, redrawn

    eng, precip, pet, resampled = make_workload(args.workload, N, dtype, rank)
    T = precip.shape[0]
    out = {"runoff": torch.empty(T, N, dtype=dtype, device=dev), "percolation": torch.empty(T, N, dtype=dtype, device=dev)}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        eng.reset()
        if i is not None:
            ev[i][0].record()  # same stream as the kernel launch (torch's current stream)
        res = eng.forward(precip, pet, series=("runoff", "percolation"), out=out, basin=("runoff",), check=False)
        if i is not None:
            ev[i][1].record()
        basin = res["basin:runoff"]  # basin runoff per timestep [T] (fp64): lgar_basin_reduce_kernel over the stored series
        if dist_on:
            all_reduce(basin)  # the only exchange of the path
        return basin

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.geff_wave_calls()  # reset the measurement counter
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    basin = None
    for i in range(args.steps):
        basin = step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist_on:
        all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    faulted = torch.tensor([int((eng.status != 0).sum().item())], device=dev)
    if dist_on:
        all_reduce(faulted)
    geff_waves = eng.geff_wave_calls() / max(args.steps, 1)  # per launch

    if rank == 0:
        units = N * world * T * args.steps
        b_alg = alg_bytes_per_col_step(elem, T)
        achieved = b_alg * N * T / (kern_ms * 1e-3) / 1e9
        # the kernel that carries the work: the 8-slot member of the front-capacity chain for jobs above 1024 waves
        # (include/lgar.h), named as rocprofv3 prints it
        kname = "lgar_forward_kernel<%s, 3, %d, 1>" % ("float" if elem == 4 else "double",
                                                        lg._capi.CAP_SMALL if N > 65536 else lg._capi.FMAX)
        traffic, traffic_note = measured_traffic(N, T, args.dtype)
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
                       "columns_per_gpu": N, "timesteps_per_pass": T, "columns_redrawn_to_stay_in_reference_domain": resampled,
                       "domain_check": "every timed column runs to the end without a fault in fp32 and in fp64 (untimed set-up)",
                       "parallelism": "columns sharded x%d" % world,
                       "collective": None if not dist_on else ("%s all-reduce of basin runoff [T], group of %d%s" % (
                           backend, world, " (LGAR_FORCE_DIST=1)" if world == 1 else ""))},
            # what binds the kernel is vector-ALU issue (~1e3 flop per algorithmic byte); achieved / peak / frac stay the
            # HBM figures the contract asks for -- algorithmic bytes per second over the 8 TB/s peak, 3 % by construction
            "roofline": {"bound": "valu-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": b_alg * N * T,
                         "kernel": kname, "kernel_ms": kern_ms,
                         "kernel_ms_covers": "HIP events around one lgar_forward call: the dominant kernel, the two ~5 us kernels of its "
                                             "capacity chain and the ~0.13 ms basin pass over the stored runoff series",
                         "alg_bytes_per_column_timestep": b_alg,
                         "note": "bound names what limits the kernel -- vector-ALU issue (~1e3 flop/B; valu_busy_frac, "
                                 "valu_roofline, DESIGN.md section 5) --; achieved / peak / frac are HBM GB/s: algorithmic bytes "
                                 "(SURVEY 8(d), F_MAX=16 state) per second of kernel time over the 8 TB/s peak"},
            # fraction of the kernel's time its vector ALU is busy (PMC, committed profile of this workload and library build)
            "valu_busy_frac": None,
            "faulted_columns": int(faulted.item()),
            "basin_runoff_total_cm": float(basin.sum().item()),
        }
        pmc = measured_valu(N, T, args.dtype)
        if pmc:
            line["valu_busy_frac"] = pmc.get("busy_frac")
        probe = {}
        if world == 1 and not args.no_extras:
            # compute-side roof, live: what this chip's vector ALU sustains vs what the Geff trapezoid (the dominant
            # instruction stream: 5 v_log/v_exp per node, 121 nodes per call) got out of it
            probe = valu_probe_record(dev)
            vr = {"bound": "valu-issue", "unit": "wave-instructions/s",
                  "peak_v_exp_f32": probe["v_exp_f32"], "peak_v_log_f32": probe["v_log_f32"],
                  "peak_v_fma_f32": probe["v_fma_f32"], "peak_geff_mix": probe["geff_mix"],
                  "definition": "peaks measured live by lgar_valu_probe (64-instruction unrolled inline-asm loops, best of 2 "
                                "and 4 waves/SIMD); geff_mix = the lean Geff loop's own instruction stream (per node pair: 4 v_log, "
                                "4 v_exp, 10 packed)"}
            if geff_waves is not None:
                nint = 120
                trans = geff_waves * ((nint + 1) * 4 + 9)  # wave-instructions: 4 per node + the two h(Se) pows and K_r(Se=1)
                vr["geff_wave_calls_per_launch"] = geff_waves
                vr["achieved_geff_transcendentals"] = trans / (kern_ms * 1e-3)
                peak_t = 0.5 * (probe["v_exp_f32"] + probe["v_log_f32"])
                vr["frac_of_transcendental_peak"] = trans / (kern_ms * 1e-3) / peak_t if peak_t else None
                vr["note"] = ("frac_of_transcendental_peak: the Geff trapezoid's v_log/v_exp wave-instructions per second of the "
                              "WHOLE kernel time over the chip's measured transcendental issue rate -- the hard floor of this "
                              "algorithm (4 transcendentals per node) is that fraction of today's time")
            if pmc:
                vr["pmc"] = pmc
            line["valu_roofline"] = vr
        elif pmc:
            line["valu_roofline"] = pmc
        if world == 1 and not args.no_extras and args.workload == "synth1":
            subs = {}
            # the parity-bearing precision on the same workload (1e-6 vs the reference holds in fp64)
            if args.dtype == "f32":
                e64, p64, q64, _ = make_workload("synth1", N, torch.float64, rank)
                e64.geff_wave_calls()  # reset the counter
                ms = []
                for _ in range(3):
                    e64.reset()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    e64.forward(p64, q64, series=("runoff", "percolation"), basin=("runoff",), check=False)
                    b.record()
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
                ms = min(ms[1:])
                subs["fp64"] = {"value": N * T / (ms * 1e-3), "unit": "column-timesteps/s", "kernel_ms": ms, "columns": N,
                                "timesteps": T, "dtype": "f64", "workload": "same synth_1 ensemble, fp64 (parity precision)",
                                "roofline_frac_hbm": alg_bytes_per_col_step(8, T) * N * T / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                # compute side of the parity precision: the fused fp64 trapezoid node is 93 vector instructions
                # (profiles/*/isa_census.txt: 2 log2 + 2 exp2 by polynomial, no hardware transcendental), all of them at the
                # v_fma_f64 issue rate except two v_rcp_f64 at a quarter of it (profiles/r02/valu_probe_f64_helpers.jsonl)
                calls64 = e64.geff_wave_calls() / 3.0
                # one node = 93 vector instructions (profiles/*/isa_census.txt), all issuing like v_fma_f64 except two v_rcp_f64
                # at a quarter of that rate (profiles/r02/valu_probe_f64_helpers.jsonl): 99 issue slots.  Peak: the chip's fp64
                # vector rate, 78.6 TFLOP/s (AMD's MI355X specification; = one 64-lane v_fma_f64 per 4 cycles per SIMD at the
                # 2.4 GHz of MI355X_MICROARCH.md) = 6.14e11 wave-instructions/s (128 flop each); the live probe
                # of v_fma_f64 is reported next to it (its own loop reaches ~75 % of that: a lower bound, not the roof).
                peak64 = 78.6e12 / 128.0
                slots = calls64 * 121 * 99
                subs["fp64"]["valu_roofline"] = {
                    "bound": "valu-issue", "unit": "wave-instructions/s", "peak_v_fma_f64": peak64,
                    "probe_v_fma_f64": probe.get("v_fma_f64"), "geff_wave_calls_per_launch": calls64,
                    "vector_instructions_per_trapezoid_node": 93, "issue_slots_per_trapezoid_node": 99,
                    "achieved_geff_issue_slots": slots / (ms * 1e-3), "frac_of_issue_peak": slots / (ms * 1e-3) / peak64,
                    "note": "the trapezoid's fp64 instruction stream, in v_fma_f64 issue slots per second of the WHOLE kernel time, "
                            "over the chip's fp64 vector issue rate: the rest of the kernel time is the column physics outside "
                            "the trapezoid and lanes idled by divergence"}
                del e64, p64, q64
                # fp64 column state with the fp32 hardware transcendentals inside the Geff trapezoid (LgarDims.geff_mode = 1):
                # run totals of every column within 2e-6 of the native fp64 kernels', identical fault flags (tests/test_gpu_mixed.py)
                emx, pmx, qmx, _ = make_workload("synth1", N, torch.float64, rank, geff_precision="f32")
                ms = []
                for _ in range(3):
                    emx.reset()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    emx.forward(pmx, qmx, series=("runoff", "percolation"), basin=("runoff",), check=False)
                    b.record()
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
                ms = min(ms[1:])
                subs["fp64_state_f32_geff"] = {
                    "value": N * T / (ms * 1e-3), "unit": "column-timesteps/s", "kernel_ms": ms, "columns": N, "timesteps": T,
                    "dtype": "f64 state / f32 trapezoid nodes",
                    "workload": "same synth_1 ensemble, mixed precision (geff_precision='f32')",
                    "faulted_columns": int((emx.status != 0).sum().item()),
                    "roofline_frac_hbm": alg_bytes_per_col_step(8, T) * N * T / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                del emx, pmx, qmx
            ec, pc, qc, _ = make_workload("phillipsburg", 10_000, torch.float64, 0)
            Tc = pc.shape[0]
            ms = []
            for _ in range(2):
                ec.reset()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ec.forward(pc, qc, series=("runoff", "percolation"), check=False)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            subs["configs1"] = {"value": 10_000 * Tc / (min(ms) * 1e-3), "unit": "column-timesteps/s", "kernel_ms": min(ms),
                                "columns": 10_000, "timesteps": Tc, "dtype": "f64",
                                "lanes_per_column": ec.cooperating_lanes(),
                                "workload": "BASELINE configs[1]: 10k replicated Phillipsburg columns x 3000 hourly steps, fp64 "
                                            "(157 waves of columns on 1024 SIMDs: bound by the time ONE wave needs for 3000 steps, "
                                            "not a throughput figure; the library gives every column the cooperating lanes that "
                                            "keep the job at one wave per SIMD -- 6 here, 150 -> 82 ms)"}
            del ec
            # ... and in the mixed-precision mode (fp64 state, fp32-transcendental trapezoid nodes): its four-node groups are split
            # over the column's lanes as well (bit for bit the mixed mode with one lane per column)
            em, _, _, _ = make_workload("phillipsburg", 10_000, torch.float64, 0, geff_precision="f32")
            ms = []
            for _ in range(2):
                em.reset()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                em.forward(pc, qc, series=("runoff", "percolation"), check=False)
                b.record()
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            subs["configs1"]["mixed_precision"] = {"kernel_ms": min(ms), "value": 10_000 * Tc / (min(ms) * 1e-3),
                                                   "lanes_per_column": em.cooperating_lanes(),
                                                   "dtype": "f64 state / f32 trapezoid nodes"}
            del em, pc, qc
            # the reference's own use: ONE column (agents/DifferentiableLGAR.py:117-125), 3000 hourly rows in one launch, with
            # one lane and with the library's choice of cooperating lanes (64 for a job this small; same results bit for bit)
            one = {}
            for label, lanes, prec in (("one_lane", 1, "native"), ("cooperating_lanes", 0, "native"),
                                       ("cooperating_lanes_mixed_precision", 0, "f32")):
                g1 = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
                P1 = W.PHILLIPSBURG
                e1 = lg.LgarEngine(*[P1[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")], n_columns=1,
                                   dt_h=1.0, ponded_depth_max=2.0, dtype=torch.float64, device=dev, forward_lanes=lanes,
                                   geff_precision=prec)
                p1 = torch.tensor(g1["forcing"][:, 0:1], device=dev).contiguous()
                q1 = torch.tensor(g1["forcing"][:, 1:2], device=dev).contiguous()
                ms = []
                for _ in range(3):
                    e1.reset()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    e1.forward(p1, q1, series=("runoff", "percolation"), check=False)
                    b.record()
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
                one[label] = {"kernel_ms": min(ms), "column_timesteps_per_s": p1.shape[0] / (min(ms) * 1e-3)}
            subs["single_column"] = dict(one, columns=1, timesteps=int(p1.shape[0]), dtype="f64",
                                         workload="the bundled Phillipsburg column x 3000 hourly steps in one launch (the "
                                                  "reference: 12.9 steps/s); jobs under one wave per SIMD give every column "
                                                  "4..64 lanes that split the Geff trapezoid and the front sweep")
            # BASELINE configs[4]: 100 000-column vG parameter ensemble, forward + backward (autograd through the HIP
            # kernels: 9 parameter directions as one tangent launch), loss = mean of runoff^2
            from lgar_py_amd.autograd import lgar_series
            Ne = 100_000
            E = W.ensemble_columns(Ne, seed=0)
            fe = W.synth1_forcing()
            Te = fe.shape[0]
            pe_ = torch.tensor(fe[:, 0:1], device=dev).expand(Te, Ne).contiguous().to(torch.float64)
            qe_ = torch.zeros_like(pe_)
            Pe = {k: torch.tensor(v, device=dev, dtype=torch.float64) for k, v in E.items()}
            for k in ("alpha", "n", "ksat"):
                Pe[k].requires_grad_(True)
            best = None
            for _ in range(2):
                for k in ("alpha", "n", "ksat"):
                    Pe[k].grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = []
                ro, _ = lgar_series(Pe["alpha"], Pe["n"], Pe["ksat"], Pe["theta_e"], Pe["theta_r"], Pe["thickness"], pe_, qe_,
                                    dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float64, check=False, status_out=st)
                okc = st[0] == 0
                loss = torch.mean(ro[:, okc] ** 2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                loss.backward()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if best is None or t2 - t0 < best[0]:
                    best = (t2 - t0, t1 - t0, t2 - t1, float(okc.double().mean()), float((st[1] != 0).double().mean()))
            subs["configs4_autograd"] = {"value": Ne * Te / best[0], "unit": "column-timesteps/s (forward + backward)",
                                         "forward_ms": 1e3 * best[1], "backward_ms": 1e3 * best[2], "columns": Ne, "timesteps": Te,
                                         "dtype": "f64", "valid_fraction": best[3], "tangent_faulted_fraction": best[4],
                                         "workload": "BASELINE configs[4]: 100k-column (alpha, n, Ksat) ensemble, loss = mean "
                                                     "runoff^2, gradients of 9 parameters per column through the HIP tangent kernel"}
            # Per-column forcing that lives on the HOST (a sharded job with real per-catchment series): file -> pinned double
            # buffer (pread by 8 threads) -> HBM on a side stream -> kernels, chunk by chunk (pipeline.run_streamed_columns);
            # then the same with the runoff series streamed BACK into a host file on a third stream, and with the input map
            # registered with the runtime (no staging copy).  These legs include the host link by construction -- they are what
            # the link sustains, never the headline value.
            try:
                import shutil
                import tempfile

                from lgar_py_amd.pipeline import (close_forcing_file, create_forcing_file, open_forcing_file, run_streamed_columns,
                                                  write_forcing_file)
                Ns, tile = 1 << 20, 1
                fs = W.synth1_forcing(tile)
                Ts = fs.shape[0]
                Ps = W.perturbed_columns(Ns, seed=7)
                scs = W.forcing_scale(Ns, 0.5, 1.0, seed=8).astype(np.float32)
                # the files go to tmpfs only when tmpfs has room for them with a margin (a store into a sparse file on a full
                # tmpfs is a SIGBUS, which no `except` catches); else to the ordinary temporary directory
                need = Ns * Ts * 4
                tmp_root = None
                if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * need + (1 << 30):
                    tmp_root = "/dev/shm"
                tmpd = tempfile.mkdtemp(prefix="lgar_bench_", dir=tmp_root)
                path, opath = os.path.join(tmpd, "precip.npy"), os.path.join(tmpd, "runoff.npy")
                try:
                    write_forcing_file(path, (fs[:, 0:1].astype(np.float32) * scs[None, :]))
                    es = lg.LgarEngine(Ps["alpha"], Ps["n"], Ps["ksat"], Ps["theta_e"], Ps["theta_r"], Ps["thickness"],
                                       dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float32, device=dev)

                    def leg(src, threads=12, chunk=16, **kw):  # best of three: the copying threads share the host with everything else
                        best_ = None
                        for _ in range(3):
                            es.reset()
                            st_ = {}
                            run_streamed_columns(es, src, None, chunk=chunk, series=("runoff",), check=False, stats=st_,
                                                 reader_threads=threads, **kw)
                            if best_ is None or st_["wall_s"] < best_["wall_s"]:
                                best_ = st_
                        return best_
                    mm = open_forcing_file(path)
                    staged = leg(mm)                                   # pread by 12 threads into the pinned double buffer
                    del mm
                    t_reg_in = time.perf_counter()
                    rm = open_forcing_file(path, register=True)        # the map registered with the runtime: no staging copy
                    t_reg_in = time.perf_counter() - t_reg_in
                    best = leg(rm, threads=1)
                    create_forcing_file(opath, (Ts, Ns), "float32")
                    t_reg = time.perf_counter()
                    om = open_forcing_file(opath, register=True)
                    t_reg = time.perf_counter() - t_reg
                    both = leg(rm, threads=1, host_out={"runoff": om})  # ... and the runoff series streamed back into a host file
                    close_forcing_file(rm)
                    close_forcing_file(om)
                    del rm, om, es
                finally:  # the files are tmpfs, i.e. memory: they go whatever happens
                    for p_ in (path, opath):
                        if os.path.exists(p_):
                            os.remove(p_)
                    os.rmdir(tmpd)
                subs["streamed_host_forcing"] = {
                    "value": best["column_timesteps_per_s"], "unit": "column-timesteps/s (host -> device included)",
                    "host_to_device_GBps": best["host_to_device_GBps"], "bytes_host_to_device": best["bytes_host_to_device"],
                    "wall_ms": 1e3 * best["wall_s"], "columns": Ns, "timesteps": Ts, "chunk_rows": best["chunk_rows"],
                    "source": best["source"], "register_ms": 1e3 * t_reg_in, "dtype": "f32",
                    "kernel_appetite_GBps_at_1e10": best["kernel_appetite_GBps_at_1e10"],
                    "pread_staging": {
                        "value": staged["column_timesteps_per_s"], "wall_ms": 1e3 * staged["wall_s"],
                        "host_to_device_GBps": staged["host_to_device_GBps"], "reader_threads": staged["reader_threads"],
                        "source": staged["source"],
                        "note": "the same job with nothing prepared: 12 threads pread the file into a pinned double buffer (every "
                                "byte crosses the host's memory three times instead of once)"},
                    "with_runoff_streamed_back": {
                        "value": both["column_timesteps_per_s"], "wall_ms": 1e3 * both["wall_s"],
                        "host_to_device_GBps": both["host_to_device_GBps"], "device_to_host_GBps": both["device_to_host_GBps"],
                        "bytes_device_to_host": both["bytes_device_to_host"],
                        "destination": "a [T, N] file on tmpfs, mapped and registered with the runtime (%.0f ms): the copy engines "
                                       "write the page cache; the link carries 57 GB/s both ways TOGETHER" % (1e3 * t_reg)},
                    "workload": "the headline's own ensemble with its [T, N] precipitation (N distinct columns, PET zero) in a file on "
                                "tmpfs: the map is registered with the runtime (hipHostRegister, register_ms, once per file, outside "
                                "the timed region), the copy engines read the page cache chunk by chunk of 16 rows on a side stream "
                                "while lgar_forward integrates the chunk before; bound by the host link (57 GB/s measured alone, "
                                "profiles/r05/hostlink.json) and by the kernels' own appetite (4 B per column-timestep), whichever is "
                                "less"}
            except Exception as e:  # noqa: BLE001 -- a sub-record must never take the headline line down
                subs["streamed_host_forcing"] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["sub_records"] = subs
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
