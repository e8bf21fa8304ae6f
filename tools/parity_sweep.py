"""Large seeded parity sweep of the HIP fp64 path against the CPU oracle (dev tool, GPU box): many more columns than the
test-suite cases, several seeds, both ensemble shapes and all three search modes.  Prints per case the fault-flag agreement
and the worst relative error over the columns both sides integrate; exits 1 on a disagreement above 1e-6 (for the
wide-parameter ensemble: on more than 0.1 % of the columns taking another branch, see the comment below).
usage: python tools/parity_sweep.py [columns] [seeds]"""
import json, os, sys, time
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
        for mode in (1, 0, 2):
            eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=kw["dt_h"],
                                ponded_depth_max=kw["pdm"], dtype=torch.float64, search_mode=mode)
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
            col_err = np.abs(got_ro - ro).max(0) / max(1.0, np.abs(ro).max())
            n_div = int((col_err[both] > 1e-6).sum())
            rec = dict(seed=seed, shape=shape, mode=mode, columns=int(pr.shape[1]), steps=int(pr.shape[0]), valid=float(both.mean()),
                       flag_mismatches=n_flag_diff, columns_off_by_more_than_1e_6=n_div, err_runoff=e_ro, err_perc=e_pc,
                       err_totals=e_tot, err_volume=e_vol, oracle_s=round(t_or, 1))
            print(json.dumps(rec), flush=True)
            worst = max(worst, e_ro, e_pc, e_tot, e_vol)
            # The wide ensemble (n up to 3, Ksat over 2.5 decades) saturates its top layers, where psi -> 0 and the
            # reference's own decisions (isclose ties at 1e-8, the Se > 1 fault of insert_water) are decided by the last
            # ulp of pow: a few columns in 10^4 take the other branch under ANY other libm or search order.  Reported,
            # not failed; the +-10 % shapes must agree exactly.
            if shape != "wide" and (n_flag_diff or max(e_ro, e_pc, e_tot, e_vol) > 1e-6):
                fail = True
            if shape == "wide" and (n_flag_diff + n_div) > 1e-3 * pr.shape[1]:
                fail = True
                bad = np.nonzero((st != 0) != (gst != 0))[0][:5]
                print("  mismatching flags at columns", bad.tolist(), "oracle", st[bad].tolist(), "hip", gst[bad].tolist(), flush=True)
print(json.dumps(dict(worst=worst, failed=fail)))
sys.exit(1 if fail else 0)
