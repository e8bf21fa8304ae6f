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
