"""CPU: the oracle (oracle/lgar_oracle.c) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; tolerance 1e-9 relative on every
per-step accumulator (observed: bit-identical on most cases, <= 3e-11 on the rest)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, check_fault_kind, golden_names
from oracle import lgar_oracle as O

RTOL = 1e-9


def _params(g):
    p = O.make_params(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"],
                         pdm=float(g["pdm"]), dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]),
                         initial_psi=float(g["initial_psi"]), wp_psi=float(g["wilting_point_psi"]),
                         frozen_factor=float(g["frozen_factor"]), nint=int(g["nint"]), giuh=g["giuh_ordinates"])
    p.closed_form = int(bool(g["closed_form"])) if "closed_form" in g.files else 0
    return p


def _rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-6)


@pytest.mark.parametrize("name", golden_names())
def test_trajectory(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    p = _params(g)
    s = O.init_state(p)
    assert abs(s.ending_volume - float(g["init_volume"])) <= 1e-12
    frec = min(g["fronts"].shape[1], O.FMAX)  # front slots recorded per step (40 in the many-front fixtures; oracle holds 32)
    r = O.run(p, s, g["forcing"][:T, 0], g["forcing"][:T, 1], frec=frec)
    assert r["status"] == 0
    assert _rel(r["acc"], g["acc"][:T]).max() <= RTOL
    assert (r["nfronts"] == g["nfronts"][:T]).all()
    assert (r["front_layer"] == g["front_layer"][:T, :frec]).all()
    assert (r["front_bottom"] == g["front_bottom"][:T, :frec]).all()
    assert _rel(r["fronts"], g["fronts"][:T, :frec]).max() <= 1e-6
    if crash >= 0:
        # the reference raised at this step (e.g. ValueError: negative pow base); the oracle must flag the same step
        r2 = O.run(p, s, g["forcing"][T:T + 1, 0], g["forcing"][T:T + 1, 1])
        assert r2["status"] != 0, str(g["crash_msg"])
        check_fault_kind(g, r2["status"])


@pytest.mark.parametrize("name", ["phil_hourly_3000", "synth0_phil_1500", "synth1_phil", "bushland_hourly_1500",
                                  "generic_phil_forcing_1000"])
def test_single_step_transitions(name):
    """Start from the reference's captured state at step k and take ONE forward(): covers every branch
    (create / insert_water / move in layer 0 and deeper / base case / merge / layer crossing / dry-over-wet)
    as a (state_in, forcing) -> state_out pair."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = _params(g)
    nf = g["nfronts"]
    T = len(nf)
    change = [k for k in range(T - 1) if nf[k + 1] != nf[k]]
    ks = sorted(set(change + list(range(0, T - 1, 37))))
    for k in ks:
        s = O.init_state(p)
        O.set_fronts(s, g["fronts"][k], g["front_layer"][k], g["front_bottom"][k], nf[k])
        s.ponded_water = float(g["acc"][k, 8])
        s.ending_volume = float(g["acc"][k, 9])
        s.previous_precip = float(g["prev_precip"][k])
        for i in range(5):
            s.giuh_queue[i] = float(g["giuh_queue"][k, i])
        r = O.run(p, s, g["forcing"][k + 1:k + 2, 0], g["forcing"][k + 1:k + 2, 1])
        assert r["status"] == 0, (name, k)
        assert _rel(r["acc"][0], g["acc"][k + 1]).max() <= RTOL, (name, k)
        assert r["nfronts"][0] == nf[k + 1], (name, k)
        assert _rel(r["fronts"][0], g["fronts"][k + 1]).max() <= 1e-6, (name, k)


def test_reference_anchor_totals():
    """Run totals the survey observed from the reference (SURVEY.md §8c 'sanity anchors')."""
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    tot = dict(zip(O.ACC_NAMES[:8], g["totals"]))
    assert abs(tot["precip"] - 19.786600000000007) < 1e-9
    assert abs(tot["AET"] - 14.240699989889816) < 1e-9
    assert abs(g["acc"][-1, 9] - 50.66175036442332) < 1e-9
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    tot = dict(zip(O.ACC_NAMES[:8], g["totals"]))
    assert abs(tot["infiltration"] - 11.79631670191385) < 1e-9
    assert abs(tot["runoff"] - 0.7036832980861533) < 1e-9


def test_leaf_kats():
    g = np.load(os.path.join(GOLDEN, "leaf_kats.npz"))
    L = O.lib()
    import ctypes as C
    st = C.c_int(0)
    for si, (al, n, ks, te, tr) in enumerate(g["soils"]):
        m = 1.0 - 1.0 / n
        for hi, h in enumerate(g["hs"]):
            assert abs(L.lgo_theta_from_h(h, al, m, n, te, tr, C.byref(st)) - g["theta_from_h"][si, hi]) <= 1e-15
            assert abs(L.lgo_se_from_h(h, al, m, n, C.byref(st)) - g["se_from_h"][si, hi]) <= 1e-15
        for ei, se in enumerate(g["ses"]):
            assert _rel(L.lgo_k_from_se(se, ks, m, C.byref(st)), g["k_from_se"][si, ei]) <= 1e-12
            assert _rel(L.lgo_h_from_se(se, al, m, n, C.byref(st)), g["h_from_se"][si, ei]) <= 1e-12
        for pi in range(g["geff"].shape[1]):
            got = L.lgo_geff(g["geff_theta1"][si, pi], g["geff_theta2"][si, pi], al, n, m, ks, te, tr, 120, C.byref(st))
            assert _rel(got, g["geff"][si, pi]) <= 1e-12
        for pi, psi in enumerate(g["aet_psis"]):
            for qi, pet in enumerate(g["aet_pets"]):
                for di, dt in enumerate(g["aet_dts"]):
                    got = L.lgo_aet(pet, dt, psi, al, n, m, te, tr, 15495.0, C.byref(st))
                    assert abs(got - g["aet"][si, pi, qi, di]) <= 1e-15
    q = (C.c_double * 16)()
    o = (C.c_double * 16)(0.06, 0.51, 0.28, 0.12, 0.03)
    for i, r in enumerate(g["giuh_runoff_in"]):
        now = L.lgo_giuh(q, o, 5, float(r))
        assert abs(now - g["giuh_out"][i]) <= 1e-15
        assert np.allclose([q[j] for j in range(5)], g["giuh_queue"][i], rtol=0, atol=1e-15)


def test_domain_bottom_sets_status():
    """The reference crashes (AttributeError, Layer.py:980) when a front reaches the domain bottom; the
    oracle reports LGO_ST_BOTTOM instead (percolation parity is unpinned: SURVEY §8c)."""
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    p = O.make_params(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], [4.0, 4.0, 4.0], pdm=0.0,
                      dt_h=float(g["dt_h"]))
    s = O.init_state(p)
    r = O.run(p, s, g["forcing"][:, 0], g["forcing"][:, 1])
    assert r["status"] & O.ST_BOTTOM


def test_many_columns_matches_single():
    g = np.load(os.path.join(GOLDEN, "synth1_pert1.npz"))
    N = 5
    rep = lambda a: np.repeat(np.asarray(a)[:, None], N, 1)
    T = g["forcing"].shape[0]
    ro, pc, acc, st = O.run_columns(rep(g["alpha"]), rep(g["n"]), rep(g["ksat"]), rep(g["theta_e"]), rep(g["theta_r"]),
                                    rep(g["thickness"]), np.repeat(g["forcing"][:, 0:1], N, 1),
                                    np.repeat(g["forcing"][:, 1:2], N, 1), pdm=float(g["pdm"]), dt_h=float(g["dt_h"]))
    assert (st == 0).all()
    assert _rel(ro[:, N - 1], g["acc"][:, 4]).max() <= RTOL
    assert _rel(acc[3], g["acc"][:, 3].sum()).max() <= RTOL
