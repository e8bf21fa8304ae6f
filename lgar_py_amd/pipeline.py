"""Streaming driver for long runs (forcing I/O at scale).

A basin forcing series x[T, 2] (what the reference's Data yields row by row, data/Data.py:32-37) times a per-column
scale is expanded to the kernels' [Tc, N] layout chunk by chunk on a side HIP stream while the main stream
integrates the previous chunk: two device buffers, no [T, N] array ever exists, and the host -> HBM copy of the
(tiny) basin series goes through pinned memory.  8760 hourly steps x 1M columns would otherwise need 2 x 35 GB.
"""
import torch

from .distributed import all_reduce_sum


def run_streamed(engine, x, scale=None, pet_scale=None, chunk=512, series=("runoff",), reduce_basin=True, weights=None,
                 check=True):
    """Integrate engine over the whole series x[T, 2] (cm/h; precip, PET).

    scale / pet_scale: optional [N] per-column forcing multipliers; weights: optional [N] basin weights (area fractions).  Returns {name: [T] basin sums (fp64, all-reduced across
    ranks when torch.distributed is initialised)} if reduce_basin else {name: [T, N]} (only sensible for small N)."""
    dev, dt = engine.device, engine.dtype
    N = engine.N
    x = torch.as_tensor(x, dtype=torch.float64)
    T = x.shape[0]
    xh = x.to(dt).contiguous().pin_memory() if dev.type == "cuda" else x.to(dt)
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    ones = torch.ones(N, dtype=dt, device=dev)
    s_p = ones if scale is None else torch.as_tensor(scale).to(dev, dt)
    s_e = ones if pet_scale is None else torch.as_tensor(pet_scale).to(dev, dt)
    bufs = [(torch.empty(chunk, N, dtype=dt, device=dev), torch.empty(chunk, N, dtype=dt, device=dev)) for _ in range(2)]
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    freed = [torch.cuda.Event(), torch.cuda.Event()]
    outs = {nm: [] for nm in series}

    def stage(ci, lo, hi):
        b = ci % 2
        with torch.cuda.stream(side):
            side.wait_event(freed[b])  # the kernel that last read this buffer has finished
            xd = xh[lo:hi].to(dev, non_blocking=True)
            torch.mul(xd[:, 0:1], s_p[None, :], out=bufs[b][0][: hi - lo])
            torch.mul(xd[:, 1:2], s_e[None, :], out=bufs[b][1][: hi - lo])
            ready[b].record(side)

    for b in range(2):
        freed[b].record(main)
    bounds = [(lo, min(lo + chunk, T)) for lo in range(0, T, chunk)]
    if bounds:
        stage(0, *bounds[0])
    for ci, (lo, hi) in enumerate(bounds):
        if ci + 1 < len(bounds):
            stage(ci + 1, *bounds[ci + 1])  # expand the next chunk while this one is integrated
        b = ci % 2
        main.wait_event(ready[b])
        if reduce_basin:
            # basin sums come out of the kernel epilogue: no [Tc, N] series buffer is written at all
            out = engine.forward(bufs[b][0][: hi - lo], bufs[b][1][: hi - lo], series=(), basin=series, weights=weights,
                                 check=False)
        else:
            out = engine.forward(bufs[b][0][: hi - lo], bufs[b][1][: hi - lo], series=series, check=False)
        freed[b].record(main)
        for nm in series:
            outs[nm].append(all_reduce_sum(out["basin:" + nm]) if reduce_basin else out[nm])
    if check:
        engine.check_status()  # raises like the reference does when a column left its domain of validity
    return {nm: (torch.cat(v) if v else torch.zeros(0, dtype=torch.float64, device=dev)) for nm, v in outs.items()}


# ------------------------------------------------------------------------------------------------------------------
# Per-column forcing from the host: [T, N] arrays that do not fit (or do not live) in HBM
# ------------------------------------------------------------------------------------------------------------------
def write_forcing_file(path, array):
    """Store a [T, N] forcing array (cm/h) as a raw row-major file that open_forcing_file maps back without reading it."""
    import numpy as np
    a = np.ascontiguousarray(array)
    if a.ndim != 2 or a.dtype not in (np.float32, np.float64):
        raise ValueError("forcing files hold [T, N] float32 / float64 arrays")
    mm = np.lib.format.open_memmap(path, mode="w+", dtype=a.dtype, shape=a.shape)
    mm[:] = a
    mm.flush()
    del mm
    return path


def forcing_csvs_to_files(csv_paths, precip_path, pet_path, nsteps=None, dtype="float32", mm_to_cm=0.1):
    """One forcing file per column in the reference's on-disk formats (`Time,P(mm/h),PET(mm/h)` CSV or the `#Time` variant of
    the synth files: data.read_forcing, data/Data.py:32-37) -> the two row-major [T, N] files run_streamed_columns maps
    (cm/h): column c is csv_paths[c].  The files are written column by column through a memory map, so N x T never has to fit
    in memory; every file must hold at least `nsteps` rows (default: the length of the first).  Returns (T, N)."""
    import numpy as np

    from .data import read_forcing
    if not csv_paths:
        raise ValueError("no forcing files given")
    _, x0 = read_forcing(csv_paths[0], nsteps, mm_to_cm)
    T, N = x0.shape[0], len(csv_paths)
    outs = [np.lib.format.open_memmap(p_, mode="w+", dtype=np.dtype(dtype), shape=(T, N)) for p_ in (precip_path, pet_path)]
    for c, path in enumerate(csv_paths):
        x = x0 if c == 0 else read_forcing(path, T, mm_to_cm)[1]
        if x.shape[0] < T:
            raise ValueError("%s holds %d forcing rows, %d are needed" % (path, x.shape[0], T))
        outs[0][:, c] = x[:T, 0]
        outs[1][:, c] = x[:T, 1]
    for o in outs:
        o.flush()
    del outs
    return T, N


def open_forcing_file(path):
    """Memory-map a file written by write_forcing_file (numpy .npy container, row-major [T, N]): pages are read when a chunk
    is staged, never the whole file."""
    import numpy as np
    mm = np.load(path, mmap_mode="r")
    if mm.ndim != 2:
        raise ValueError("%s does not hold a [T, N] array" % path)
    return mm


def run_streamed_columns(engine, precip, pet=None, chunk=256, series=("runoff",), reduce_basin=True, weights=None, check=True,
                         stats=None, reader_threads=4):
    """Integrate engine over a forcing series with N DISTINCT columns that lives on the host -- what a sharded job with real
    per-catchment forcing has (the reference's Data yields ONE basin series row by row, data/Data.py:32-37; run_streamed above
    is that case).  precip: [T, N] host array (numpy, typically open_forcing_file's memory map; cm/h, float32 or float64);
    pet: the same, or [T] / [T, 1] (one basin series, expanded on the device), or None (zero).

    Three stages run concurrently, two buffers each:
      file pages -> pinned host buffer   a reader thread that splits every chunk's rows over `reader_threads` copying threads
                                         (numpy's copy loops release the GIL; one core moves ~12 GB/s out of the page cache,
                                         less than the link carries: the page faults of the map happen here),
      pinned -> HBM                      cudaMemcpyAsync on a side stream (+ the dtype conversion, if the file's differs),
      kernels                            lgar_forward on the caller's stream over the chunk that has landed.
    No [T, N] array ever exists on the device.  Returns what run_streamed returns.  stats (a dict, optional) receives the
    bytes moved, the wall time and the host -> device rate that was reached end to end (the kernels' appetite at 1e10
    column-timesteps/s is 8 bytes per column-timestep in fp32: 87 GB/s, more than a PCIe 5.0 x16 link carries -- a job of
    distinct columns streamed from the host is bound by that link, not by the kernels)."""
    import queue
    import threading
    import time

    import numpy as np
    dev, dt = engine.device, engine.dtype
    N = engine.N
    if precip.ndim != 2 or precip.shape[1] != N:
        raise ValueError("precip must be [T, %d]; got %s" % (N, tuple(precip.shape)))
    T = precip.shape[0]
    pet_kind = "none" if pet is None else ("full" if (getattr(pet, "ndim", 0) == 2 and pet.shape[1] == N and N != 1) else "basin")
    if pet_kind == "full" and tuple(pet.shape) != (T, N):
        raise ValueError("pet must be [T, N], [T] or None")
    if pet_kind == "basin":
        pet = np.asarray(pet).reshape(-1)
        if pet.shape[0] != T:
            raise ValueError("pet must be [T, N], [T] or None")
    np_dt = {torch.float32: np.float32, torch.float64: np.float64}
    src_dt = torch.float32 if precip.dtype == np.float32 else torch.float64
    n_full = 2 if pet_kind == "full" else 1
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    pinned = [torch.empty(n_full, chunk, N, dtype=src_dt).pin_memory() for _ in range(2)]
    pinned_pet = [torch.empty(chunk, 1, dtype=torch.float64).pin_memory() for _ in range(2)] if pet_kind == "basin" else None
    staged = [torch.empty(n_full, chunk, N, dtype=src_dt, device=dev) for _ in range(2)] if src_dt != dt else None
    bufs = [(torch.empty(chunk, N, dtype=dt, device=dev), torch.zeros(chunk, N, dtype=dt, device=dev)) for _ in range(2)]
    copied = [torch.cuda.Event(), torch.cuda.Event()]  # the H2D copy out of pinned buffer b has finished
    ready = [torch.cuda.Event(), torch.cuda.Event()]   # device buffer b holds its chunk
    freed = [torch.cuda.Event(), torch.cuda.Event()]   # the kernels that read device buffer b have finished
    bounds = [(lo, min(lo + chunk, T)) for lo in range(0, T, chunk)]
    filled = queue.Queue(maxsize=1)  # chunk indices whose pinned buffer is full
    pinned_free = [threading.Semaphore(1), threading.Semaphore(1)]
    err = []

    from concurrent.futures import ThreadPoolExecutor
    nthr = max(1, int(reader_threads))
    pool = ThreadPoolExecutor(max_workers=nthr) if nthr > 1 else None

    def copy_rows(dst, src, lo, hi):
        """dst[: hi - lo] = src[lo:hi], the rows split over the copying threads"""
        n = hi - lo
        if pool is None or n < 2 * nthr:
            np.copyto(dst[:n], src[lo:hi])
            return
        cuts = [lo + (n * k) // nthr for k in range(nthr + 1)]
        futs = [pool.submit(np.copyto, dst[a - lo:b_ - lo], src[a:b_]) for a, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a]
        for f in futs:
            f.result()

    stop = threading.Event()  # set when the consumer gives up (an exception below): the reader must not wait for ever

    def reader():
        try:
            for ci, (lo, hi) in enumerate(bounds):
                b = ci % 2
                while not pinned_free[b].acquire(timeout=0.2):  # the copy that last read this pinned buffer has finished
                    if stop.is_set():
                        return
                if stop.is_set():
                    return
                h = pinned[b].numpy()
                copy_rows(h[0], precip, lo, hi)
                if pet_kind == "full":
                    copy_rows(h[1], pet, lo, hi)
                elif pet_kind == "basin":
                    pinned_pet[b].numpy()[: hi - lo, 0] = pet[lo:hi]
                while True:
                    try:
                        filled.put(ci, timeout=0.2)
                        break
                    except queue.Full:
                        if stop.is_set():
                            return
        except Exception as e:  # noqa: BLE001 -- handed to the consumer, which re-raises
            err.append(e)
            try:
                filled.put(-1, timeout=1.0)
            except queue.Full:
                pass

    th = threading.Thread(target=reader, daemon=True)
    outs = {nm: [] for nm in series}
    for b in range(2):
        freed[b].record(main)
    t0 = time.perf_counter()
    th.start()
    pending_release = []  # (event, pinned index): released to the reader once the copy has finished

    def upload(ci):
        while True:
            try:
                got = filled.get(timeout=0.5)
                break
            except queue.Empty:
                if not th.is_alive():
                    raise (err[0] if err else RuntimeError("the forcing reader ended before chunk %d" % ci))
        if got < 0:
            raise err[0]
        lo, hi = bounds[ci]
        b, n = ci % 2, hi - lo
        with torch.cuda.stream(side):
            side.wait_event(freed[b])
            if staged is None:
                bufs[b][0][:n].copy_(pinned[b][0, :n], non_blocking=True)
                if pet_kind == "full":
                    bufs[b][1][:n].copy_(pinned[b][1, :n], non_blocking=True)
            else:  # the file's precision differs from the engine's: convert on the device, not on the host
                staged[b][:, :n].copy_(pinned[b][:, :n], non_blocking=True)
                bufs[b][0][:n].copy_(staged[b][0, :n])
                if pet_kind == "full":
                    bufs[b][1][:n].copy_(staged[b][1, :n])
            if pet_kind == "basin":
                col = pinned_pet[b][:n].to(dev, non_blocking=True)
                bufs[b][1][:n] = col.to(dt)  # [n, 1] broadcast over the columns
            copied[b].record(side)
            ready[b].record(side)
        pending_release.append((copied[b], b))

    def release_finished(block=False):
        while pending_release and (block or pending_release[0][0].query()):
            ev, b = pending_release.pop(0)
            ev.synchronize()
            pinned_free[b].release()

    try:
        if bounds:
            upload(0)
        for ci, (lo, hi) in enumerate(bounds):
            if ci + 1 < len(bounds):
                release_finished(block=True)  # the reader may refill the pinned buffer the previous upload has drained
                upload(ci + 1)
            b, n = ci % 2, hi - lo
            main.wait_event(ready[b])
            if reduce_basin:
                out = engine.forward(bufs[b][0][:n], bufs[b][1][:n], series=(), basin=series, weights=weights, check=False)
            else:
                out = engine.forward(bufs[b][0][:n], bufs[b][1][:n], series=series, check=False)
            freed[b].record(main)
            for nm in series:
                outs[nm].append(all_reduce_sum(out["basin:" + nm]) if reduce_basin else out[nm])
        release_finished(block=True)
    finally:  # whatever happened: the reader and its copying threads end, and nothing still reads the pinned buffers
        stop.set()
        th.join()
        if pool is not None:
            pool.shutdown(wait=True)
        torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if stats is not None:
        moved = T * N * n_full * (4 if src_dt == torch.float32 else 8)
        stats.update(bytes_host_to_device=moved, wall_s=wall, host_to_device_GBps=moved / wall / 1e9,
                     column_timesteps_per_s=T * N / wall, chunks=len(bounds), chunk_rows=chunk, reader_threads=nthr,
                     kernel_appetite_GBps_at_1e10=1e10 * 2 * (4 if dt == torch.float32 else 8) / 1e9)
    if check:
        engine.check_status()
    return {nm: (torch.cat(v) if v else torch.zeros(0, dtype=torch.float64, device=dev)) for nm, v in outs.items()}
