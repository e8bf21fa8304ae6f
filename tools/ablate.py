"""Cost attribution of the forward kernel (dev tool).

`python tools/ablate.py build` (CPU container: hipcc cross-compiles) builds measurement variants of liblgar_hip.so:
duplication variants (-DLGAR_DUP_<X>: routine X runs twice, results unchanged => time delta = cost of X), skip variants
and occupancy variants.  `python tools/ablate.py run [f32|f64|mix] [columns]` (GPU box) times the bench workload's forward
launch on each variant in a fresh child process and prints one JSON line per variant.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "base": [],
    "dup_geff": ["-DLGAR_DUP_GEFF"],
    "dup_search": ["-DLGAR_DUP_SEARCH"],
    "dup_mb": ["-DLGAR_DUP_MB"],
    "dup_fdd": ["-DLGAR_DUP_FDD"],
    "dup_event": ["-DLGAR_DUP_EVENT"],
    "dup_dzdt": ["-DLGAR_DUP_DZDT"],
    "nogeff": ["-DLGAR_ABL_NOGEFF"],
    "nogeff_nodzdt": ["-DLGAR_ABL_NOGEFF", "-DLGAR_ABL_NODZDT"],
    "nogeff_nomove": ["-DLGAR_ABL_NOGEFF", "-DLGAR_ABL_NOMOVE"],
    "nogeff_noinsert": ["-DLGAR_ABL_NOGEFF", "-DLGAR_ABL_NOINSERT"],
    "nogeff_nosearch": ["-DLGAR_ABL_NOGEFF", "-DLGAR_ABL_NOSEARCH"],
    "clocks": ["-DLGAR_CLOCKS"],  # cycle attribution of a one-wave job: tools/smalljob_clocks.py
    "skeleton": ["-DLGAR_ABL_NOGEFF", "-DLGAR_ABL_NODZDT", "-DLGAR_ABL_NOMOVE", "-DLGAR_ABL_NOINSERT"],
    "contract": ["-ffp-contract=fast"],
    "count_lanes": ["-DLGAR_COUNT_LANES"],  # geff_calls then counts LANE-level evaluations (base: wave-level)
    "site1_waves": ["-DLGAR_COUNT_SITE=1"], "site1_lanes": ["-DLGAR_COUNT_SITE=1", "-DLGAR_COUNT_LANES"],  # calc_dzdt
    "site1_le16": ["-DLGAR_COUNT_SITE=1", "-DLGAR_COUNT_MAXLANES=16"], "site1_le32": ["-DLGAR_COUNT_SITE=1", "-DLGAR_COUNT_MAXLANES=32"],
    "site1_le8": ["-DLGAR_COUNT_SITE=1", "-DLGAR_COUNT_MAXLANES=8"], "site1_le48": ["-DLGAR_COUNT_SITE=1", "-DLGAR_COUNT_MAXLANES=48"],
    "site2_waves": ["-DLGAR_COUNT_SITE=2"], "site2_lanes": ["-DLGAR_COUNT_SITE=2", "-DLGAR_COUNT_LANES"],  # dry depth
    "site3_waves": ["-DLGAR_COUNT_SITE=3"], "site3_lanes": ["-DLGAR_COUNT_SITE=3", "-DLGAR_COUNT_LANES"],  # insert_water
    "cf_o2": ["-O2"], "cf_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "cf_relaxocc": ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=true"], "cf_nopostsched": ["-mllvm", "-enable-post-misched=0"],
    "cf_nounroll": ["-fno-unroll-loops"], "cf_metricbias": ["-mllvm", "-amdgpu-schedule-metric-bias=30"], "cf_trackers": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
    "cf_iterative": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"], "cf_nohighrp": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
    "tan_clocks": ["-DLGAR_CLOCKS"],  # cycle attribution of the tangent kernel: tools/tangent_clocks.py
    "tan_base": [], "tan_maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "tan_f32w1": ["-DLGAR_TAN_F32_WAVES=1"],
    # mixed-precision kernels only (quick to build): experiments on geff_mixed, run with `ablate.py run mix`
    "mx_base": ["-DLGAR_ONLY_MIXED"], "mx_general": ["-DLGAR_ONLY_MIXED", "-DLGAR_GEFFM_GENERAL_ONLY"],
    "mx_regions": ["-DLGAR_ONLY_MIXED", "-DLGAR_COUNT_GEFFM_REGIONS"],
    "f32_only": ["-DLGAR_ONLY_F32"], 
    "mx_occ1": ["-DLGAR_ONLY_MIXED", "-DLGAR_OCC_F64_SMALL=1"],
    "mx_occ1_ilp": ["-DLGAR_ONLY_MIXED", "-DLGAR_OCC_F64_SMALL=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "mx_ilp": ["-DLGAR_ONLY_MIXED", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "occ3": ["-DLGAR_OCC_F32_SMALL=3"],
    "occ2": ["-DLGAR_OCC_F32_SMALL=2"],
    "occ1_f64": ["-DLGAR_OCC_F64_SMALL=1"],
}

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
dtype = torch.float32 if %(dt)r == "f32" else torch.float64
kw = {"geff_precision": "f32"} if %(dt)r == "mix" else {}
N = %(n)d
f = W.synth1_forcing(); T = f.shape[0]
P = W.perturbed_columns(N, seed=0)
sc = torch.tensor(W.forcing_scale(N, seed=1000), device="cuda")
eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                    ponded_depth_max=0.0, dtype=dtype, **kw)
pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * sc[None, :]).to(dtype).contiguous()
pe = torch.zeros_like(pr)
out = {"runoff": torch.empty(T, N, dtype=dtype, device="cuda"), "percolation": torch.empty(T, N, dtype=dtype, device="cuda")}
ms = []
for rep in range(%(reps)d):
    eng.reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.forward(pr, pe, series=("runoff", "percolation"), out=out, check=False)
    b.record(); torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
ms = sorted(ms[1:])
dbg = None
if hasattr(eng.lib, "lgar_debug_counters"):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    if eng.lib.lgar_debug_counters(buf, 1) == 0:
        dbg = list(buf)
print(json.dumps(dict(variant=%(name)r, dtype=%(dt)r, columns=N, ms_median=ms[len(ms) // 2], ms_min=ms[0], dbg=dbg,
                      col_steps_per_s=N * T / (ms[len(ms) // 2] * 1e-3), faulted=int((eng.status != 0).sum()),
                      geff_calls=int(eng.geff_wave_calls()), runoff_sum=float(out["runoff"].double().sum()))))
"""


def main():
    from lgar_py_amd import build as B
    what = sys.argv[1] if len(sys.argv) > 1 else "build"
    names = [v for v in os.environ.get("LGAR_VARIANTS", "").split(",") if v] or list(VARIANTS)
    if what == "build":
        for nm in names:
            print(nm, B.build_variant(nm, VARIANTS[nm], tangent=nm.startswith("tan_")), flush=True)
        return
    dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
    for nm in names:
        lib = os.path.join(B.CSRC, "variants", "liblgar_hip_%s.so" % nm)
        if not os.path.exists(lib):
            print(json.dumps(dict(variant=nm, error="not built")), flush=True)
            continue
        env = dict(os.environ, LGAR_LIB=lib)
        code = CHILD % dict(root=ROOT, dt=dt, n=n, reps=6, name=nm)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        print(p.stdout.strip() or json.dumps(dict(variant=nm, error=p.stderr[-400:])), flush=True)


if __name__ == "__main__":
    main()
