"""Per-column forcing streamed from a file (pipeline.run_streamed_columns): host -> device rate and end-to-end column-timesteps/s per
number of reader threads and chunk size, for the three ways a chunk leaves the host (pread into pinned buffers; numpy copies
out of a plain array; a registered map, no staging) and with the runoff series streamed back.  (dev tool)
usage: bench_streamed_columns.py [N] [tile] [chunk,chunk,...]"""
import json, os, shutil, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
from lgar_py_amd.pipeline import close_forcing_file, create_forcing_file, open_forcing_file, run_streamed_columns, write_forcing_file
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4
CHUNKS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [24, 48, 96]
MID = CHUNKS[len(CHUNKS) // 2]
f = W.synth1_forcing(tile); T = f.shape[0]
P = W.perturbed_columns(N, seed=7); sc = W.forcing_scale(N, 0.5, 1.0, seed=8).astype(np.float32)
need = N * T * 4
shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * need + (1 << 30) else None
d = tempfile.mkdtemp(prefix="lgar_stream_", dir=shm)
path, opath = os.path.join(d, "precip.npy"), os.path.join(d, "runoff.npy")
try:
    write_forcing_file(path, f[:, 0:1].astype(np.float32) * sc[None, :])
    create_forcing_file(opath, (T, N), "float32")
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float32)

    def run(label, src, thr, chunk, **kw):
        best = None
        for _ in range(3):
            eng.reset(); st = {}
            run_streamed_columns(eng, src, None, chunk=chunk, series=("runoff",), check=False, stats=st, reader_threads=thr, **kw)
            if best is None or st["wall_s"] < best["wall_s"]: best = st
        print(json.dumps(dict(leg=label, columns=N, steps=T, reader_threads=thr, chunk_rows=chunk, wall_ms=round(1e3 * best["wall_s"], 2),
                              host_to_device_GBps=round(best["host_to_device_GBps"], 2), device_to_host_GBps=round(best["device_to_host_GBps"], 2),
                              column_timesteps_per_s=best["column_timesteps_per_s"], source=best["source"])), flush=True)
    mm = open_forcing_file(path)
    for thr in (4, 8, 12):
        for chunk in CHUNKS:
            run("pread", mm, thr, chunk)
    run("numpy copies (a plain array)", np.array(mm), 8, MID)
    om = open_forcing_file(opath, register=True)
    run("pread + runoff back (registered output map)", mm, 8, MID, host_out={"runoff": om})
    close_forcing_file(om); del om
    ow = open_forcing_file(opath, writable=True)
    run("pread + runoff back (plain output map, writer thread)", mm, 8, MID, host_out={"runoff": ow})
    del ow, mm
    rm = open_forcing_file(path, register=True)
    for chunk in CHUNKS:
        run("registered input map", rm, 1, chunk)
    close_forcing_file(rm); del rm
finally:
    for p_ in (path, opath):
        if os.path.exists(p_): os.remove(p_)
    os.rmdir(d)
