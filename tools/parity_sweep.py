"""Large seeded parity sweep of the HIP fp64 path against the CPU oracle (dev tool, GPU box): many more columns than the
test-suite cases, several seeds, both ensemble shapes and all three search modes.  Prints per case the fault-flag agreement
and the worst relative error over the columns both sides integrate; exits 1 on a disagreement above 1e-6 (for the
wide-parameter ensemble: on more than 0.1 % of the columns taking another branch, see the comment below).
usage: python tools/parity_sweep.py [columns] [seeds]"""
import json, os, sys, time, zlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lgar_py_amd as lg
from lgar_py_amd import workloads as W
from oracle import lgar_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = W.synth1_forcing()
worst = 0.0
fail = False
for seed in range(SEEDS):
    for shape in ("perturbed", "wide", "hourly"):
        if shape == "perturbed":
            P = W.perturbed_columns(N, seed=100 + seed); sc = W.forcing_scale(N, seed=200 + seed)
            pr = f[:, 0:1] * sc[None, :]; pe = np.zeros_like(pr); kw = dict(dt_h=300.0 / 3600.0, pdm=0.0)
        elif shape == "wide":
            P = W.ensemble_columns(N, seed=300 + seed)
            pr = np.repeat(f[:, 0:1], N, 1); pe = np.zeros_like(pr); kw = dict(dt_h=300.0 / 3600.0, pdm=0.0)
        else:
            g = np.load(os.path.join(ROOT, "tests", "golden", "phil_hourly_3000.npz"))
            n2 = N // 8
            P = W.perturbed_columns(n2, seed=400 + seed); sc = W.forcing_scale(n2, seed=500 + seed, lo=0.5, hi=3.0)
            pr = g["forcing"][:600, 0:1] * sc[None, :]; pe = np.repeat(g["forcing"][:600, 1:2], n2, 1); kw = dict(dt_h=1.0, pdm=2.0)
        t0 = time.time()
        ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                        pdm=kw["pdm"], dt_h=kw["dt_h"])
        t_or = time.time() - t0
        for mode in (1, 0, 2, "mixed"):
            # "mixed": search mode 1 with the mixed-precision trapezoid (geff_precision="f32"): identical fault flags required,
            # run totals within 2e-6 of the column's water input, per-step runoff within 2e-4 of the largest;
            # the wide-parameter shape is reported only (there 0.2 % of the columns change their fault flag in the mixed mode)
            mk = dict(search_mode=1, geff_precision="f32") if mode == "mixed" else dict(search_mode=mode)
            eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=kw["dt_h"],
                                ponded_depth_max=kw["pdm"], dtype=torch.float64, **mk)
            out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff", "percolation"), check=False)
            gst = eng.status.cpu().numpy() & 0x7f
            both = (st == 0) & (gst == 0)
            flags_equal = bool(((st != 0) == (gst != 0)).all())
            # the iteration cap of the literal searches is the one fault the oracle does not model identically
            n_flag_diff = int(((st != 0) != (gst != 0)).sum())
            got_ro = out["runoff"].cpu().numpy(); got_pc = out["percolation"].cpu().numpy()
            e_ro = float(np.abs(got_ro - ro)[:, both].max() / max(1.0, np.abs(ro).max()))
            e_pc = float(np.abs(got_pc - pc)[:, both].max() / max(1.0, np.abs(pc).max()))
            tot = eng.totals.cpu().numpy()
            e_tot = float((np.abs(tot[:8] - acc[:8]) / np.maximum(np.abs(acc[:8]), 1e-3))[:, both].max())
            e_vol = float((np.abs(tot[9] - acc[9]) / np.maximum(np.abs(acc[9]), 1e-6))[both].max())
            if mode == "mixed":  # totals against the column's water input
                e_tot = float((np.abs(tot[:8] - acc[:8]) / np.maximum(np.maximum(np.abs(acc[:8]), acc[0:1]), 1e-2))[:, both].max())
            col_err = np.abs(got_ro - ro).max(0) / max(1.0, np.abs(ro).max())
            n_div = int((col_err[both] > 1e-6).sum())
            rec = dict(seed=seed, shape=shape, mode=mode, columns=int(pr.shape[1]), steps=int(pr.shape[0]), valid=float(both.mean()),
                       flag_mismatches=n_flag_diff, columns_off_by_more_than_1e_6=n_div, err_runoff=e_ro, err_perc=e_pc,
                       err_totals=e_tot, err_volume=e_vol, oracle_s=round(t_or, 1))
            print(json.dumps(rec), flush=True)
            # The wide ensemble (n up to 3, Ksat over 2.5 decades) saturates its top layers, where psi -> 0 and the
            # reference's own decisions (isclose ties at 1e-8, the Se > 1 fault of insert_water) are decided by the last
            # ulp of pow: a few columns in 10^4 take the other branch under ANY other libm or search order.  Reported,
            # not failed; the +-10 % shapes must agree exactly.
            if mode == "mixed":
                if shape != "wide" and (n_flag_diff or e_tot > 2e-6 or e_vol > 2e-6 or max(e_ro, e_pc) > 2e-4):
                    fail = True
                continue
            worst = max(worst, e_ro, e_pc, e_tot, e_vol)
            if shape != "wide" and (n_flag_diff or max(e_ro, e_pc, e_tot, e_vol) > 1e-6):
                fail = True
            if shape == "wide" and (n_flag_diff + n_div) > 1e-3 * pr.shape[1]:
                fail = True
                bad = np.nonzero((st != 0) != (gst != 0))[0][:5]
                print("  mismatching flags at columns", bad.tolist(), "oracle", st[bad].tolist(), "hip", gst[bad].tolist(), flush=True)

# ---- second part: every configuration family of the golden fixtures (layer counts 2..6, closed-form G, frozen factor, initial
# psi, sub-cycling, ponding limits), each soil perturbed +-10 % per column, rain scaled U(0.5, 2) per column
FIXTURES = ["two_layer_synth1", "two_layer_phil_600", "four_layer_synth1", "four_layer_phil_600", "five_layer_synth1",
            "five_layer_phil_500", "six_layer_synth1", "six_layer_phil_300", "closedG_synth1_phil", "closedG_generic_synth0_400",
            "frozen07_synth1_phil", "frozen07_phil_hourly_400", "psi500_synth1_generic", "bushland_hourly_1500",
            "generic_phil_forcing_1000", "phil_5min_600h", "synth3_generic", "manyfronts_pulse_84", "rand00", "rand02", "rand05",
            "rand07", "rand09", "rand10"]
NF = int(os.environ.get("SWEEP_FIXTURE_COLUMNS", 2048))
for name in (FIXTURES if os.environ.get("SWEEP_FIXTURES", "1") != "0" else []):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    L = len(g["alpha"])
    closed = bool(g["closed_form"]) if "closed_form" in g.files else False
    n = NF // 4 if closed else NF
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    P = {k: np.asarray(g[k], dtype=np.float64)[:, None] * (1.0 + 0.10 * (2.0 * rng.random((L, n)) - 1.0))
         for k in ("alpha", "n", "ksat", "theta_e", "theta_r")}
    P["thickness"] = np.repeat(np.asarray(g["thickness"], dtype=np.float64)[:, None], n, 1)
    T = min(g["forcing"].shape[0], 400)
    sc = 0.5 + 1.5 * rng.random(n)
    pr = g["forcing"][:T, 0:1] * sc[None, :]
    pe = np.repeat(g["forcing"][:T, 1:2], n, 1)
    kw = dict(initial_psi=float(g["initial_psi"]), pdm=float(g["pdm"]), wp_psi=float(g["wilting_point_psi"]),
              frozen_factor=float(g["frozen_factor"]), dt_h=float(g["dt_h"]), nint=int(g["nint"]),
              num_subcycles=int(g["num_subcycles"]), giuh=tuple(g["giuh_ordinates"]))
    if not closed:
        ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe, **kw)
    else:
        ro = np.zeros((T, n)); acc = np.zeros((10, n)); st = np.zeros(n, dtype=np.int32)
        for c in range(n):
            pp = O.make_params(*(P[k][:, c] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")), **kw)
            pp.closed_form = 1
            ss = O.init_state(pp)
            r = O.run(pp, ss, pr[:, c], pe[:, c], fronts=False)
            ro[:, c] = r["acc"][:, 4]; acc[:, c] = r["acc"].sum(0); acc[9, c] = r["acc"][-1, 9]; st[c] = r["status"]
    for mode in ((1, 0, 2) if closed else (1, 0, 2, "mixed")):
        mk = dict(search_mode=1, geff_precision="f32") if mode == "mixed" else dict(search_mode=mode)
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=kw["dt_h"],
                            num_subcycles=kw["num_subcycles"], ponded_depth_max=kw["pdm"], initial_psi=kw["initial_psi"],
                            wilting_point_psi=kw["wp_psi"], frozen_factor=kw["frozen_factor"], nint=kw["nint"],
                            giuh_ordinates=kw["giuh"], use_closed_form_G=closed, dtype=torch.float64, **mk)
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff",), check=False)
        gst = eng.status.cpu().numpy() & 0x7f
        both = (st == 0) & (gst == 0)
        n_flag_diff = int(((st != 0) != (gst != 0)).sum())
        got = out["runoff"].cpu().numpy()
        col_err = np.abs(got - ro).max(0) / max(1.0, np.abs(ro).max())
        tot = eng.totals.cpu().numpy()
        e_tot = float((np.abs(tot[2:6] - acc[2:6]) / np.maximum(np.abs(acc[2:6]), 1e-3))[:, both].max()) if both.any() else 0.0
        if mode == "mixed" and both.any():
            e_tot = float((np.abs(tot[2:6] - acc[2:6]) / np.maximum(np.maximum(np.abs(acc[2:6]), acc[0:1]), 1e-2))[:, both].max())
        e_ro = float(col_err[both].max()) if both.any() else 0.0
        rec = dict(fixture=name, layers=L, mode=mode, columns=n, steps=T, valid=float(both.mean()), flag_mismatches=n_flag_diff,
                   columns_off_by_more_than_1e_6=int((col_err[both] > 1e-6).sum()), err_runoff=e_ro, err_totals=e_tot,
                   max_fronts=int(eng.n_fronts.max()))
        print(json.dumps(rec), flush=True)
        if mode == "mixed":
            if n_flag_diff or e_tot > 5e-6 or e_ro > 2e-4:  # (observed: 2.4e-6 on the perturbed two_layer_synth1 family, else <= 8e-7)
                fail = True
                print("  mixed mode outside its bars", flush=True)
            continue
        worst = max(worst, e_ro, e_tot)
        if n_flag_diff or max(e_ro, e_tot) > 1e-6:
            fail = True
            bad = np.nonzero(((st != 0) != (gst != 0)) | (both & (col_err > 1e-6)))[0][:6]
            print("  differing columns", bad.tolist(), "oracle", st[bad].tolist(), "hip", gst[bad].tolist(),
                  "err", [float(col_err[b]) for b in bad], flush=True)
print(json.dumps(dict(worst=worst, failed=fail)))
sys.exit(1 if fail else 0)
