"""What the host link of this box sustains, piece by piece (dev tool, GPU box): pinned -> HBM and HBM -> pinned copies, staging a
file's bytes into pinned memory (memory-map + numpy copy vs pread straight into the pinned buffer, per thread count), and DMA
straight out of a host-registered memory map of the file.  usage: python tools/hostlink_probe.py [MB]"""
import ctypes
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nbytes = MB << 20
dev = torch.device("cuda:0")
out = {"MB": MB, "cores": len(os.sched_getaffinity(0))}
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
free = None
if shm:
    import shutil
    free = shutil.disk_usage(shm).free
    out["shm_free_GB"] = round(free / 1e9, 2)
    if free < 2 * nbytes + (256 << 20):
        shm = None
tmpd = tempfile.mkdtemp(prefix="lgar_probe_", dir=shm)
path = os.path.join(tmpd, "x.bin")
out["file_on"] = "tmpfs" if shm else "disk"
try:
    a = np.random.default_rng(0).random(nbytes // 4, dtype=np.float32)
    with open(path, "wb") as fh:
        fh.write(a.tobytes())
    del a
    d = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    h = torch.empty(nbytes // 4, dtype=torch.float32).pin_memory()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return nbytes / best / 1e9

    out["pinned_to_hbm_GBps"] = timed(lambda: d.copy_(h, non_blocking=True))
    out["hbm_to_pinned_GBps"] = timed(lambda: h.copy_(d, non_blocking=True))
    # both directions at once, on two streams
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    h2 = torch.empty_like(h).pin_memory()
    d2 = torch.empty_like(d)

    def both():
        with torch.cuda.stream(s1):
            d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)
    out["bidirectional_each_GBps"] = timed(both)
    hv = h.numpy().view(np.uint8)
    fd = os.open(path, os.O_RDONLY)

    def stage_pread(nthr):
        cuts = [(nbytes * k // nthr) & ~4095 for k in range(nthr)] + [nbytes]
        def work(a_, b_):
            mv = memoryview(hv[a_:b_])
            off = a_
            while off < b_:
                n = os.preadv(fd, [mv[off - a_:]], off)
                if n <= 0:
                    raise IOError("short read")
                off += n
        ths = [threading.Thread(target=work, args=(cuts[k], cuts[k + 1])) for k in range(nthr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    mm = np.memmap(path, dtype=np.uint8, mode="r")

    def stage_mmap(nthr):
        cuts = [(nbytes * k // nthr) & ~4095 for k in range(nthr)] + [nbytes]
        ths = [threading.Thread(target=np.copyto, args=(hv[cuts[k]:cuts[k + 1]], mm[cuts[k]:cuts[k + 1]])) for k in range(nthr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    for nm, fn in (("pread", stage_pread), ("mmap_copy", stage_mmap)):
        for nthr in (1, 2, 4, 8, 12, 16):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                fn(nthr)
                best = min(best, time.perf_counter() - t0)
            out["stage_%s_%dthr_GBps" % (nm, nthr)] = nbytes / best / 1e9
    os.close(fd)
    del mm
    # DMA straight out of the page cache: a shared writable map of the file, host-registered
    rt = torch.cuda.cudart()
    mmw = np.memmap(path, dtype=np.float32, mode="r+")
    t0 = time.perf_counter()
    rc = rt.cudaHostRegister(mmw.ctypes.data, nbytes, 0)
    out["host_register_rc"] = int(rc) if not isinstance(rc, int) else rc
    out["host_register_s"] = time.perf_counter() - t0
    if int(out["host_register_rc"]) == 0:
        tm = torch.from_numpy(mmw)
        out["registered_map_is_pinned"] = bool(tm.is_pinned())
        out["registered_map_to_hbm_GBps"] = timed(lambda: d.copy_(tm, non_blocking=True))
        chk = torch.empty_like(d)
        chk.copy_(h)  # h holds the file's bytes (staged above)
        out["registered_map_copy_correct"] = bool(torch.equal(chk, d))
        del tm
        rt.cudaHostUnregister(mmw.ctypes.data)
    del mmw
finally:
    if os.path.exists(path):
        os.remove(path)
    os.rmdir(tmpd)
print(json.dumps(out, indent=1))
