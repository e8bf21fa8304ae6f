"""Per-column forcing streamed from a memory-mapped file (pipeline.run_streamed_columns): host -> device rate and end-to-end
column-timesteps/s per number of copying threads and chunk size.  (dev tool)  usage: bench_streamed_columns.py [N] [tile]"""
import json, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
from lgar_py_amd.pipeline import open_forcing_file, run_streamed_columns, write_forcing_file
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = W.synth1_forcing(tile); T = f.shape[0]
P = W.perturbed_columns(N, seed=7); sc = W.forcing_scale(N, 0.5, 1.0, seed=8).astype(np.float32)
d = tempfile.mkdtemp(prefix="lgar_stream_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "precip.npy")
write_forcing_file(path, f[:, 0:1].astype(np.float32) * sc[None, :])
mm = open_forcing_file(path)
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=torch.float32)
for thr in (1, 2, 4, 8):
    for chunk in (48, 96, 192):
        best = None
        for _ in range(2):
            eng.reset(); st = {}
            run_streamed_columns(eng, mm, None, chunk=chunk, series=("runoff",), check=False, stats=st, reader_threads=thr)
            if best is None or st["wall_s"] < best["wall_s"]: best = st
        print(json.dumps(dict(columns=N, steps=T, reader_threads=thr, chunk_rows=chunk, wall_ms=round(1e3 * best["wall_s"], 2),
                              host_to_device_GBps=round(best["host_to_device_GBps"], 2), column_timesteps_per_s=best["column_timesteps_per_s"])), flush=True)
del mm; os.remove(path); os.rmdir(d)
