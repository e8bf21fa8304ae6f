"""Mixed-precision Geff (geff_precision="f32", LgarDims.geff_mode = 1) on the GPU: accuracy on every golden trajectory, speed on
the bench ensemble next to native fp64 and fp32, agreement with native fp64 on a large ensemble, and the accuracy of the
hardware v_log_f32 / v_exp_f32 the mode rests on.  (dev tool)  usage: python tools/mixed_probe.py [N] [out.json]"""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import ACC_NAMES
from lgar_py_amd import workloads as W

G = os.path.join(ROOT, "tests", "golden")


def golden_case(name, **kw):
    g = np.load(os.path.join(G, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=2,
                        dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]), ponded_depth_max=float(g["pdm"]),
                        initial_psi=float(g["initial_psi"]), wilting_point_psi=float(g["wilting_point_psi"]),
                        frozen_factor=float(g["frozen_factor"]), nint=int(g["nint"]), giuh_ordinates=tuple(g["giuh_ordinates"]),
                        use_closed_form_G=bool(g["closed_form"]) if "closed_form" in g.files else False, dtype=torch.float64, **kw)
    f = torch.tensor(g["forcing"][:T])
    out = eng.forward(f[:, 0:1].expand(T, 2).contiguous(), f[:, 1:2].expand(T, 2).contiguous(), series=ACC_NAMES, check=False)
    acc = np.stack([out[nm][:, 0].cpu().numpy() for nm in ACC_NAMES], 1)
    ref = g["acc"][:T]
    step = (np.abs(acc - ref) / np.maximum(np.abs(ref), 1e-6)).max()
    cum = (np.abs(np.cumsum(acc[:, :8], 0) - np.cumsum(ref[:, :8], 0)) / np.maximum(np.abs(np.cumsum(ref[:, :8], 0)), 1e-3)).max()
    flux = (np.abs(acc[:, :8] - ref[:, :8]) / np.maximum(np.maximum(ref[:, 0:1], np.abs(ref[:, :8])), 1e-3)).max()
    fr = eng.fronts()
    nf = int(g["nfronts"][T - 1])
    ok = int(fr["n_fronts"][0]) == nf and int(eng.status[0]) == 0
    fz = (np.abs(fr["depth"][:nf, 0] - g["fronts"][T - 1, :nf, 0]) / np.maximum(np.abs(g["fronts"][T - 1, :nf, 0]), 1e-6)).max() if ok else np.nan
    ft = (np.abs(fr["theta"][:nf, 0] - g["fronts"][T - 1, :nf, 1]) / np.maximum(np.abs(g["fronts"][T - 1, :nf, 1]), 1e-6)).max() if ok else np.nan
    return dict(step=float(step), cum=float(cum), flux=float(flux), fronts=float(max(fz, ft)), ok=bool(ok))


def timed(eng, pr, pe, reps=3):
    best = 1e9
    for _ in range(reps):
        eng.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.forward(pr, pe, series=("runoff", "percolation"), basin=("runoff",), check=False)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    rec = {"device": torch.cuda.get_device_name(0)}
    # 1. hardware transcendental accuracy (fp32 leaf ops 7 / 8 = v_log_f32 / v_exp_f32 as the trapezoid uses them)
    one = np.ones(1 << 16)
    kw = dict(alpha=one, n=one * 2, ksat=one, theta_e=one, theta_r=one * 0, dtype=torch.float32)
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(np.log(1e-4), np.log(1e4), 1 << 16)).astype(np.float32).astype(np.float64)
    got = lg.leaf_batch("log2", x, **kw).cpu().numpy().astype(np.float64)
    rec["v_log_f32_abs_err_over_ulp_of_result"] = float(np.max(np.abs(got - np.log2(x)) / np.spacing(np.abs(np.log2(x)).astype(np.float32))))
    rec["v_log_f32_max_abs_err"] = float(np.max(np.abs(got - np.log2(x))))
    xn = (1.0 + 10 ** rng.uniform(-6, -0.3, 1 << 16)).astype(np.float32).astype(np.float64)
    got = lg.leaf_batch("log2", xn, **kw).cpu().numpy().astype(np.float64)
    rel = np.abs(got - np.log2(xn)) / np.abs(np.log2(xn))
    rec["v_log_f32_near_one"] = {"max_rel_err": float(rel.max()), "p99_rel_err": float(np.percentile(rel, 99)),
                                 "max_abs_err": float(np.max(np.abs(got - np.log2(xn))))}
    y = rng.uniform(-20, 20, 1 << 16).astype(np.float32).astype(np.float64)
    got = lg.leaf_batch("exp2", y, **kw).cpu().numpy().astype(np.float64)
    rec["v_exp_f32_max_rel_err"] = float(np.max(np.abs(got - np.exp2(y)) / np.exp2(y)))
    print(json.dumps({k: rec[k] for k in rec if k.startswith("v_")}), flush=True)
    # 2. golden trajectories, mixed vs native
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(G, "*.npz")) if "leaf" not in f and "grad_" not in f)
    gold = {}
    for nm in names:
        m = golden_case(nm, geff_precision="f32")
        n_ = golden_case(nm)
        gold[nm] = {"mixed": m, "native": n_}
        print("%-34s mixed step %.2e cum %.2e flux %.2e fronts %.2e ok %s | native step %.2e" % (
            nm, m["step"], m["cum"], m["flux"], m["fronts"], m["ok"], n_["step"]), flush=True)
    rec["golden"] = gold
    rec["golden_worst"] = {k: max(v["mixed"][k] for v in gold.values() if v["mixed"]["ok"]) for k in ("step", "cum", "flux", "fronts")}
    rec["golden_all_ok"] = all(v["mixed"]["ok"] for v in gold.values())
    print(json.dumps(rec["golden_worst"]), rec["golden_all_ok"], flush=True)
    # 3. speed and ensemble agreement
    P = W.perturbed_columns(N, seed=0)
    sc = W.forcing_scale(N, seed=1000)
    f = W.synth1_forcing()
    res = {}
    for label, dt, kw2 in (("f64", torch.float64, {}), ("mixed", torch.float64, {"geff_precision": "f32"}), ("f32", torch.float32, {})):
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=dt, **kw2)
        pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[None, :]).to(dt).contiguous()
        pe = torch.zeros_like(pr)
        ms = timed(eng, pr, pe)
        res[label] = dict(ms=ms, rate=N * f.shape[0] / (ms * 1e-3), totals=eng.totals.double().clone(), status=eng.status.clone())
        print(label, "%.2f ms  %.3e column-timesteps/s  faulted %d" % (ms, res[label]["rate"], int((eng.status != 0).sum())), flush=True)
        del eng, pr, pe
    rec["speed"] = {k: {"ms": v["ms"], "column_timesteps_per_s": v["rate"], "faulted": int((v["status"] != 0).sum())} for k, v in res.items()}
    t64, s64 = res["f64"]["totals"], res["f64"]["status"]
    for label in ("mixed", "f32"):
        t, s = res[label]["totals"], res[label]["status"]
        ok = (s64 == 0) & (s == 0)
        d = {"status_mismatch": int((s64 != s).sum()), "columns": N}
        for j, scale_row, nm in ((3, 3, "infiltration"), (4, 0, "runoff_vs_precip"), (9, 9, "ending_volume")):
            r = ((t[j] - t64[j]).abs() / torch.clamp(t64[scale_row].abs(), min=1.0))[ok]
            d[nm] = {"median": float(r.median()), "p99": float(torch.quantile(r[: 1 << 22].float(), 0.99)), "max": float(r.max())}
        rec["ensemble_vs_f64_" + label] = d
        print(label, json.dumps(d), flush=True)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
