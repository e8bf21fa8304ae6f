"""Streaming driver for long runs (forcing I/O at scale).

A basin forcing series x[T, 2] (what the reference's Data yields row by row, data/Data.py:32-37) times a per-column
scale is expanded to the kernels' [Tc, N] layout chunk by chunk on a side HIP stream while the main stream
integrates the previous chunk: two device buffers, no [T, N] array ever exists, and the host -> HBM copy of the
(tiny) basin series goes through pinned memory.  8760 hourly steps x 1M columns would otherwise need 2 x 35 GB.
"""
import os

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
    try:
        # reserve the blocks first: without room (a small tmpfs, say) this fails as OSError (ENOSPC) here, where it can be
        # caught -- a store through the map into a sparse file that cannot grow is a SIGBUS
        with open(path, "r+b") as fh:
            os.posix_fallocate(fh.fileno(), 0, mm.offset + a.nbytes)
        mm[:] = a
        mm.flush()
    except BaseException:
        del mm
        if os.path.exists(path):
            os.remove(path)
        raise
    del mm
    return path


def forcing_csvs_to_files(csv_paths, precip_path, pet_path, nsteps=None, dtype="float32", mm_to_cm=0.1, block_columns=1024):
    """One forcing file per column in the reference's on-disk formats (`Time,P(mm/h),PET(mm/h)` CSV or the `#Time` variant of
    the synth files: data.read_forcing, data/Data.py:32-37) -> the two row-major [T, N] files run_streamed_columns maps
    (cm/h): column c is csv_paths[c].  The files are written through a memory map, `block_columns` columns at a time ([T, B]
    buffers: every page of the row-major files is dirtied once per block, not once per column), so N x T never has to fit in
    memory; every file must hold at least `nsteps` rows (default: the length of the first).  The outputs appear under their
    names only when complete (written as *.part, renamed at the end, removed on failure).  Returns (T, N)."""
    import numpy as np

    from .data import read_forcing
    if not csv_paths:
        raise ValueError("no forcing files given")
    _, x0 = read_forcing(csv_paths[0], nsteps, mm_to_cm)
    T, N = x0.shape[0], len(csv_paths)
    tmp = [p_ + ".part" for p_ in (precip_path, pet_path)]
    outs = []
    try:
        for p_ in tmp:
            outs.append(np.lib.format.open_memmap(p_, mode="w+", dtype=np.dtype(dtype), shape=(T, N)))
            with open(p_, "r+b") as fh:
                os.posix_fallocate(fh.fileno(), 0, outs[-1].offset + outs[-1].nbytes)
        # a block of columns at a time: every page of the row-major files is written once per block, not once per column
        B = max(1, min(N, block_columns))
        buf = np.empty((2, T, B), dtype=np.dtype(dtype))
        for c0 in range(0, N, B):
            c1 = min(c0 + B, N)
            for c in range(c0, c1):
                x = x0 if c == 0 else read_forcing(csv_paths[c], T, mm_to_cm)[1]
                if x.shape[0] < T:
                    raise ValueError("%s holds %d forcing rows, %d are needed" % (csv_paths[c], x.shape[0], T))
                buf[0, :, c - c0] = x[:T, 0]
                buf[1, :, c - c0] = x[:T, 1]
            outs[0][:, c0:c1] = buf[0, :, : c1 - c0]
            outs[1][:, c0:c1] = buf[1, :, : c1 - c0]
        for o in outs:
            o.flush()
        del outs
        for t_, p_ in zip(tmp, (precip_path, pet_path)):  # complete files only, under their final names
            os.replace(t_, p_)
    except BaseException:
        outs = None
        for t_ in tmp:
            if os.path.exists(t_):
                os.remove(t_)
        raise
    return T, N


def open_forcing_file(path, register=False, writable=False):
    """Memory-map a file written by write_forcing_file (numpy .npy container, row-major [T, N]): pages are read when a chunk
    is staged, never the whole file.
    register=True: the map (shared, writable) is also REGISTERED with the HIP runtime (hipHostRegister): the copy engines then
    read the page cache directly -- no staging copy at all, the link's full rate (57 GB/s measured on the MI355X box) -- at
    the one-time price of pinning the file's pages (~45 ms per GB); run_streamed_columns recognises such a map.  Release it
    with close_forcing_file.  writable=True alone gives a writable map (an output file for `host_out`)."""
    import numpy as np
    mm = np.load(path, mmap_mode="r+" if (register or writable) else "r")
    if mm.ndim != 2:
        raise ValueError("%s does not hold a [T, N] array" % path)
    if register:
        rc = torch.cuda.cudart().cudaHostRegister(mm.ctypes.data, mm.nbytes, 0)
        if int(rc) != 0:
            raise RuntimeError("hipHostRegister of %s failed with code %d" % (path, int(rc)))
    return mm


def create_forcing_file(path, shape, dtype="float32"):
    """An empty [T, N] file in write_forcing_file's format with its blocks reserved (an output file for run_streamed_columns'
    `host_out`); open it with open_forcing_file(path, writable=True) or register=True."""
    import numpy as np
    mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.dtype(dtype), shape=tuple(shape))
    try:
        with open(path, "r+b") as fh:
            os.posix_fallocate(fh.fileno(), 0, mm.offset + mm.nbytes)
    except BaseException:
        del mm
        if os.path.exists(path):
            os.remove(path)
        raise
    del mm
    return path


def close_forcing_file(mm):
    """Undo open_forcing_file(register=True): unregister the map from the HIP runtime (a no-op for a plain map)."""
    try:
        if _is_registered(mm):
            torch.cuda.cudart().cudaHostUnregister(mm.ctypes.data)
    except Exception:  # noqa: BLE001 -- best effort at teardown
        pass


def _is_registered(arr):
    """Is this host array page-locked memory the copy engines can address (a registered map, a pinned tensor's array)?"""
    import numpy as np
    if not isinstance(arr, np.ndarray) or not arr.flags["C_CONTIGUOUS"] or arr.size == 0:
        return False
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (a read-only map: torch warns that the tensor must not be written -- it is not)
            return bool(torch.from_numpy(arr[:1]).is_pinned())
    except Exception:  # noqa: BLE001
        return False


def _file_extent(arr):
    """(path, byte offset of element [0, 0]) when arr is a whole-file memory map that pread can address, else None."""
    import numpy as np
    if isinstance(arr, np.memmap) and getattr(arr, "filename", None) and arr.flags["C_CONTIGUOUS"] and arr.ndim == 2:
        return str(arr.filename), int(arr.offset)
    return None


def chunk_bounds(T, chunk):
    """[(lo, hi), ...] covering rows 0..T: `chunk` rows at a time, or -- chunk a sequence -- by that SCHEDULE of row counts, whose
    last entry repeats (a short first chunk starts the kernels after a fraction of a millisecond of copying, long later ones
    keep the number of launches, each with its ramp-up and tail, small)."""
    import numpy as np
    sizes = [int(chunk)] if np.ndim(chunk) == 0 else [int(c) for c in chunk]
    if not sizes or min(sizes) < 1:
        raise ValueError("chunk must be a positive row count or a non-empty sequence of them")
    bounds, lo = [], 0
    while lo < T:
        n = sizes[min(len(bounds), len(sizes) - 1)]
        bounds.append((lo, min(lo + n, T)))
        lo += n
    return bounds


def run_streamed_columns(engine, precip, pet=None, chunk=256, series=("runoff",), reduce_basin=True, weights=None, check=True,
                         stats=None, reader_threads=8, host_out=None):
    """Integrate engine over a forcing series with N DISTINCT columns that lives on the host -- what a sharded job with real
    per-catchment forcing has (the reference's Data yields ONE basin series row by row, data/Data.py:32-37; run_streamed above
    is that case).  chunk: rows per launch, or a schedule of row counts (chunk_bounds).
    precip: [T, N] host array (numpy, typically open_forcing_file's memory map; cm/h, float32 or float64);
    pet: the same, or [T] / [T, 1] (one basin series, expanded on the device), or None (zero).

    Host -> device, two buffers per stage, every stage overlapping the others:
      source -> pinned host buffer   a reader thread; how it fills the buffer depends on the source:
                                       * a memory map of a file (open_forcing_file): `reader_threads` threads pread() disjoint
                                         byte ranges of the chunk straight into the pinned buffer (no page faults on the
                                         map; 68 GB/s with 8 threads on the MI355X box, above the link's 57 GB/s),
                                       * any other array: numpy copies, the chunk's rows split over the threads;
                                     a REGISTERED map (open_forcing_file(register=True)) skips this stage altogether: the copy
                                     engines read the page cache,
      pinned -> HBM                  cudaMemcpyAsync on a side stream (+ the dtype conversion, if the file's differs),
      kernels                        lgar_forward on the caller's stream over the chunk that has landed.
    Device -> host (host_out = {series name: [T, N] host array}, optional): every chunk's per-step series goes back on a third
    stream while the next chunk is integrated -- directly into page-locked destinations (a registered writable map, a pinned
    tensor's array), through a pinned double buffer and a writer thread otherwise.
    No [T, N] array ever exists on the device.  Returns what run_streamed returns (basin sums, or with reduce_basin=False the
    series themselves on the device).  stats (a dict, optional) receives the bytes moved each way, the wall time and the
    rates reached end to end (the kernels' appetite at 1e10 column-timesteps/s is 8 bytes per column-timestep in fp32:
    80 GB/s, more than the link carries -- a job of distinct columns streamed from the host is bound by that link)."""
    import queue
    import threading
    import time

    import numpy as np
    dev, dt = engine.device, engine.dtype
    N = engine.N
    if precip.ndim != 2 or precip.shape[1] != N:
        raise ValueError("precip must be [T, %d]; got %s" % (N, tuple(precip.shape)))
    T = precip.shape[0]
    bounds = chunk_bounds(T, chunk)  # (chunk: rows per launch, or a schedule of them)
    chunk = max((hi_ - lo_ for lo_, hi_ in bounds), default=1)  # rows of every staging buffer
    pet_kind = "none" if pet is None else ("full" if (getattr(pet, "ndim", 0) == 2 and pet.shape[1] == N and N != 1) else "basin")
    if pet_kind == "full" and tuple(pet.shape) != (T, N):
        raise ValueError("pet must be [T, N], [T] or None")
    if pet_kind == "basin":
        pet = np.asarray(pet).reshape(-1)
        if pet.shape[0] != T:
            raise ValueError("pet must be [T, N], [T] or None")
    host_out = dict(host_out or {})
    for nm, dst in host_out.items():
        if tuple(dst.shape) != (T, N):
            raise ValueError("host_out[%r] must be [T, N]" % nm)
    src_dt = torch.float32 if precip.dtype == np.float32 else torch.float64
    sources = [precip] + ([pet] if pet_kind == "full" else [])
    n_full = len(sources)
    # how the chunks leave the host: straight out of page-locked memory, by pread into a pinned buffer, or by numpy copies
    direct = all(_is_registered(a) for a in sources)
    extents = [None if direct else _file_extent(a) for a in sources]
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    back = torch.cuda.Stream(dev) if host_out else None
    pinned = None if direct else [torch.empty(n_full, chunk, N, dtype=src_dt).pin_memory() for _ in range(2)]
    pinned_pet = [torch.empty(chunk, 1, dtype=torch.float64).pin_memory() for _ in range(2)] if pet_kind == "basin" else None
    staged = [torch.empty(n_full, chunk, N, dtype=src_dt, device=dev) for _ in range(2)] if src_dt != dt else None
    bufs = [(torch.empty(chunk, N, dtype=dt, device=dev), torch.zeros(chunk, N, dtype=dt, device=dev)) for _ in range(2)]
    stored = tuple(host_out) if (reduce_basin or not host_out) else tuple(dict.fromkeys(tuple(series) + tuple(host_out)))
    out_bufs = [{nm: torch.empty(chunk, N, dtype=dt, device=dev) for nm in host_out} for _ in range(2)] if host_out else None
    copied = [torch.cuda.Event(), torch.cuda.Event()]  # the H2D copy out of pinned buffer b has finished
    ready = [torch.cuda.Event(), torch.cuda.Event()]   # device buffer b holds its chunk
    freed = [torch.cuda.Event(), torch.cuda.Event()]   # the kernels that read device buffer b have finished
    landed = [torch.cuda.Event(), torch.cuda.Event()]  # the D2H copies out of output buffer b have finished
    filled = queue.Queue(maxsize=1)  # chunk indices whose pinned buffer is full
    enqueued = [threading.Event() for _ in bounds]  # the H2D copy of chunk ci has been put on the side stream
    err = []
    stop = threading.Event()  # set when the consumer gives up (an exception below): the helpers must not wait for ever

    from concurrent.futures import ThreadPoolExecutor
    nthr = max(1, int(reader_threads))
    pool = ThreadPoolExecutor(max_workers=nthr) if nthr > 1 else None
    fds = [os.open(e[0], os.O_RDONLY) if e is not None else None for e in extents]

    def fill(dst, src, fd, extent, lo, hi):
        """dst[: hi - lo] = src[lo:hi] (dst: rows of a pinned buffer), the work split over the copying threads"""
        n = hi - lo
        if extent is not None:  # a file map: pread disjoint byte ranges straight into the pinned rows
            row = N * dst.itemsize
            flat = dst[:n].reshape(-1).view(np.uint8)
            base = extent[1] + lo * row
            total = n * row
            cuts = [(total * k // nthr) & ~4095 for k in range(nthr)] + [total]

            def rd(a, b_):
                mv = memoryview(flat[a:b_])
                off = 0
                while off < b_ - a:
                    got = os.preadv(fd, [mv[off:]], base + a + off)
                    if got <= 0:
                        raise OSError("short read of the forcing file at byte %d" % (base + a + off))
                    off += got
            jobs = [(a, b_) for a, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a]
            if pool is None or len(jobs) < 2:
                for a, b_ in jobs:
                    rd(a, b_)
            else:
                for f_ in [pool.submit(rd, a, b_) for a, b_ in jobs]:
                    f_.result()
            return
        if pool is None or n < 2 * nthr:
            np.copyto(dst[:n], src[lo:hi])
            return
        cuts = [lo + (n * k) // nthr for k in range(nthr + 1)]
        for f_ in [pool.submit(np.copyto, dst[a - lo:b_ - lo], src[a:b_]) for a, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a]:
            f_.result()

    def reader():
        try:
            for ci, (lo, hi) in enumerate(bounds):
                b = ci % 2
                if ci >= 2:  # the copy that last read this pinned buffer (chunk ci - 2) must have finished
                    while not enqueued[ci - 2].wait(timeout=0.2):
                        if stop.is_set():
                            return
                    copied[b].synchronize()
                if stop.is_set():
                    return
                h = pinned[b].numpy()
                for k, a in enumerate(sources):
                    fill(h[k], a, fds[k], extents[k], lo, hi)
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

    th = threading.Thread(target=reader, daemon=True) if not direct else None
    # device -> host through a pinned double buffer for destinations that are not page-locked
    out_direct = {nm: _is_registered(dst) for nm, dst in host_out.items()}
    out_pinned = [{nm: torch.empty(chunk, N, dtype=dt).pin_memory() for nm, d_ in out_direct.items() if not d_} for _ in range(2)] \
        if host_out and not all(out_direct.values()) else None
    to_write = queue.Queue()
    out_free = [threading.Semaphore(1), threading.Semaphore(1)]

    def writer():
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                ci, ev = item
                lo, hi = bounds[ci]
                b, n = ci % 2, hi - lo
                ev.synchronize()
                for nm, t_ in out_pinned[b].items():
                    src = t_.numpy()[:n]
                    dst = host_out[nm]
                    if pool is None or n < 2 * nthr:
                        np.copyto(dst[lo:hi], src)
                    else:
                        cuts = [(n * k) // nthr for k in range(nthr + 1)]
                        for f_ in [pool.submit(np.copyto, dst[lo + a:lo + b_], src[a:b_]) for a, b_ in zip(cuts[:-1], cuts[1:]) if b_ > a]:
                            f_.result()
                out_free[b].release()
        except Exception as e:  # noqa: BLE001
            err.append(e)
            for sem in out_free:
                sem.release()

    wt = threading.Thread(target=writer, daemon=True) if out_pinned is not None else None
    outs = {nm: [] for nm in series}
    for b in range(2):
        freed[b].record(main)
        landed[b].record(main)
    t0 = time.perf_counter()
    if th is not None:
        th.start()
    if wt is not None:
        wt.start()

    def upload(ci):
        lo, hi = bounds[ci]
        b, n = ci % 2, hi - lo
        if not direct:
            while True:
                try:
                    got = filled.get(timeout=0.5)
                    break
                except queue.Empty:
                    if not th.is_alive():
                        raise (err[0] if err else RuntimeError("the forcing reader ended before chunk %d" % ci))
            if got < 0:
                raise err[0]
            host = [pinned[b][k, :n] for k in range(n_full)]
        else:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # (read-only maps: the tensors are only read)
                host = [torch.from_numpy(a[lo:hi]) for a in sources]
        if pet_kind == "basin":
            pinned_pet[b].numpy()[:n, 0] = pet[lo:hi]
        with torch.cuda.stream(side):
            side.wait_event(freed[b])
            if staged is None:
                bufs[b][0][:n].copy_(host[0], non_blocking=True)
                if pet_kind == "full":
                    bufs[b][1][:n].copy_(host[1], non_blocking=True)
            else:  # the file's precision differs from the engine's: convert on the device, not on the host
                for k in range(n_full):
                    staged[b][k, :n].copy_(host[k], non_blocking=True)
                bufs[b][0][:n].copy_(staged[b][0, :n])
                if pet_kind == "full":
                    bufs[b][1][:n].copy_(staged[b][1, :n])
            if pet_kind == "basin":
                col = pinned_pet[b][:n].to(dev, non_blocking=True)
                bufs[b][1][:n] = col.to(dt)  # [n, 1] broadcast over the columns
            copied[b].record(side)
            ready[b].record(side)
        enqueued[ci].set()

    try:
        if bounds:
            upload(0)
        for ci, (lo, hi) in enumerate(bounds):
            b, n = ci % 2, hi - lo
            main.wait_event(ready[b])
            kw = {}
            if host_out:
                main.wait_event(landed[b])  # the copies that last read output buffer b have finished
                kw["out"] = {nm: t_[:n] for nm, t_ in out_bufs[b].items()}
            if reduce_basin:
                out = engine.forward(bufs[b][0][:n], bufs[b][1][:n], series=stored, basin=series, weights=weights, check=False, **kw)
            else:
                out = engine.forward(bufs[b][0][:n], bufs[b][1][:n], series=stored if host_out else series, check=False, **kw)
            freed[b].record(main)
            if host_out:
                if out_pinned is not None:
                    out_free[b].acquire()  # the writer has emptied pinned output buffer b
                    if err:
                        raise err[0]
                with torch.cuda.stream(back):
                    back.wait_event(freed[b])
                    import warnings
                    for nm, dst in host_out.items():
                        if out_direct[nm]:
                            with warnings.catch_warnings():
                                warnings.simplefilter("ignore")
                                torch.from_numpy(dst[lo:hi]).copy_(out[nm], non_blocking=True)
                        else:
                            out_pinned[b][nm][:n].copy_(out[nm], non_blocking=True)
                    landed[b].record(back)
                if out_pinned is not None:
                    ev = torch.cuda.Event()
                    ev.record(back)
                    to_write.put((ci, ev))
            for nm in series:
                if reduce_basin:
                    outs[nm].append(all_reduce_sum(out["basin:" + nm]))
                else:
                    outs[nm].append(out[nm].clone() if host_out and nm in host_out else out[nm])
            if ci + 1 < len(bounds):
                upload(ci + 1)  # (after this chunk's kernels are on their way: the reader's wait does not hold them back)
        if wt is not None:
            to_write.put(None)
            wt.join()
            if err:
                raise err[0]
    finally:  # whatever happened: the helpers end, and nothing still reads or writes the pinned buffers
        stop.set()
        for e_ in enqueued:
            e_.set()
        if th is not None:
            th.join()
        if wt is not None and wt.is_alive():
            to_write.put(None)
            for sem in out_free:
                sem.release()
            wt.join()
        if pool is not None:
            pool.shutdown(wait=True)
        for fd in fds:
            if fd is not None:
                os.close(fd)
        torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if stats is not None:
        es = 4 if src_dt == torch.float32 else 8
        moved = T * N * n_full * es
        back_bytes = T * N * len(host_out) * (4 if dt == torch.float32 else 8)
        stats.update(bytes_host_to_device=moved, wall_s=wall, host_to_device_GBps=moved / wall / 1e9,
                     bytes_device_to_host=back_bytes, device_to_host_GBps=back_bytes / wall / 1e9,
                     column_timesteps_per_s=T * N / wall, chunks=len(bounds), chunk_rows=chunk,
                     chunk_schedule=[h_ - l_ for l_, h_ in bounds], reader_threads=nthr,
                     source="registered map (no staging)" if direct else ("pread into pinned buffers" if all(e is not None for e in extents)
                                                                           else "numpy copies into pinned buffers"),
                     kernel_appetite_GBps_at_1e10=1e10 * 2 * (4 if dt == torch.float32 else 8) / 1e9)
    if check:
        engine.check_status()
    return {nm: (torch.cat(v) if v else torch.zeros(0, dtype=torch.float64, device=dev)) for nm, v in outs.items()}
