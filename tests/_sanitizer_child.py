"""Child process of tests/test_sanitizers.py: runs under LD_PRELOAD of a sanitizer runtime and loads the AddressSanitizer +
UBSan build of the oracle (`oracle`) or of the device-code simulator (`devsim`).  Any finding aborts the process
(-fno-sanitize-recover=all); a clean run prints SANITIZER_CHILD_OK.  TEST INFRASTRUCTURE ONLY."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]
GOLDEN = os.path.join(HERE, "golden")
CRASH = ["crash_bottom_thin444_synth1", "crash_bottom_three_layer_synth3", "crash_dry_over_wet_300", "crash_insert_water_bench_col2"]
TRAJ = ["synth1_phil", "bench_col15731", "manyfronts_pulse_84", "phil_pert2_500", "frozen07_synth1_phil", "closedG_synth1_phil"]
GRAD = ["grad_synth0_12h", "grad_manyfronts_60"]


def steps_of(g, crash_too=True):
    crash = int(g["crash_step"])
    T = g["forcing"].shape[0] if crash < 0 else crash + (1 if crash_too else 0)
    return g["forcing"][:T]


def run_oracle():
    from oracle import lgar_oracle as O
    assert O.build().endswith("_asan.so")
    for name in CRASH + TRAJ:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        p = O.make_params(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], pdm=float(g["pdm"]),
                          dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]), initial_psi=float(g["initial_psi"]),
                          wp_psi=float(g["wilting_point_psi"]), frozen_factor=float(g["frozen_factor"]), nint=int(g["nint"]),
                          giuh=g["giuh_ordinates"])
        p.closed_form = int(bool(g["closed_form"])) if "closed_form" in g.files else 0
        if len(g["alpha"]) != 3:
            continue
        f = steps_of(g)
        r = O.run(p, O.init_state(p), f[:, 0], f[:, 1], frec=8)
        assert (r["status"] != 0) == (int(g["crash_step"]) >= 0), name
    # the OpenMP many-column entry point (what bench.py's cpu_baseline times)
    from lgar_py_amd import workloads as W
    P = W.perturbed_columns(64, seed=0)
    f = W.synth1_forcing()
    O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                  f[:, 0:1] * W.forcing_scale(64, seed=1000)[None, :], np.zeros((f.shape[0], 64)), dt_h=300.0 / 3600.0, pdm=0.0)


def run_devsim():
    import devsim
    assert os.environ.get("DEVSIM_SANITIZE") == "1"

    def engine(g, **kw):
        return devsim.SimEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=1,
                                dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]), ponded_depth_max=float(g["pdm"]),
                                initial_psi=float(g["initial_psi"]), wilting_point_psi=float(g["wilting_point_psi"]),
                                frozen_factor=float(g["frozen_factor"]), nint=int(g["nint"]),
                                giuh_ordinates=tuple(g["giuh_ordinates"]),
                                use_closed_form_G=bool(g["closed_form"]) if "closed_form" in g.files else False, **kw)

    # (search_mode, geff_mode, dtype): literal, fast, fast through the capacity chain, mixed precision, single precision
    modes = [(0, 0, np.float64), (1, 0, np.float64), (2, 0, np.float64), (2, 1, np.float64), (2, 0, np.float32)]
    for name in CRASH + TRAJ:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        if len(g["alpha"]) != 3:
            continue
        f = steps_of(g)
        if name == "phil_pert2_500":
            f = f[:200]
        for sm, gm, dt in modes:
            eng = engine(g, search_mode=sm, geff_mode=gm, dtype=dt)
            eng.forward(f[:, 0:1], f[:, 1:2], series=devsim.ACC_NAMES, call_sums=True)
            if dt == np.float64:
                assert (int(eng.status[0]) != 0) == (int(g["crash_step"]) >= 0), (name, sm, gm)
    for name in GRAD:  # the tangent kernels' lane body (dual numbers)
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        if len(g["alpha"]) != 3:
            continue
        f = g["forcing"]
        for sm in (1, 2):
            eng = engine(g, search_mode=sm)
            L = len(g["alpha"])
            for key in ("alpha", "n", "ksat"):
                for k in range(L):
                    d = {key: np.eye(L)[:, k:k + 1]}
                    eng.tangent(d, f[:, 0:1], f[:, 1:2], w_runoff=np.ones((f.shape[0], 1)))


if __name__ == "__main__":
    {"oracle": run_oracle, "devsim": run_devsim}[sys.argv[1]]()
    print("SANITIZER_CHILD_OK", flush=True)
