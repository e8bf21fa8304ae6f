"""VALU issue-rate probe (dev tool, GPU box): instructions/s per kind at 1..4 resident waves per SIMD.
Prints one JSON line per (op, waves/SIMD): wave-instructions per SIMD-cycle needs the clock, so both inst/s and
the implied cycles per wave-instruction at the nominal 2.4 GHz are given."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lgar_py_amd import _capi

OPS = ["exp", "log", "rcp", "sqrt", "fma", "mul", "pk_fma", "pk_mul", "cndmask", "cmp", "exp_dep", "fma_dep", "fma64",
       "mul64", "add64", "rcp64", "geff_mix", "cndmask_sgpr", "bfi", "cmp_cndmask", "add", "readlane", "ds_read", "min",
       "ldexp64", "frexp_exp64", "rndne64", "cvt_i32_f64", "cvt_f64_i32", "add_u32"]


def probe(op, waves_per_simd, iters=2000, rounds=4, n_cu=256):
    lib = _capi.load()
    lds = (160 * 1024) // (4 * waves_per_simd) if waves_per_simd < 8 else 0
    lds = (lds // 256) * 256
    nwg = n_cu * 4 * waves_per_simd * rounds
    sink = torch.zeros(nwg * 64, dtype=torch.float32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = None
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.lgar_valu_probe(OPS.index(op), nwg, lds, iters, sink.data_ptr(), st)
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        ms = a.elapsed_time(b)
        best = ms if best is None or ms < best else best
    wave_insts = nwg * iters * lib.lgar_valu_probe_insts(OPS.index(op))
    per_s = wave_insts / (best * 1e-3)
    return dict(op=op, waves_per_simd=waves_per_simd, ms=best, wave_insts_per_s=per_s,
                lane_ops_per_s=per_s * 64, cycles_per_wave_inst_per_simd_at_2p4GHz=2.4e9 * n_cu * 4 / per_s)


if __name__ == "__main__":
    ops = sys.argv[1].split(",") if len(sys.argv) > 1 else OPS
    for op in ops:
        for w in (1, 2, 3, 4):
            print(json.dumps(probe(op, w)), flush=True)
