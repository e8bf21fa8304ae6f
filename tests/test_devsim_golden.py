"""CPU: the DEVICE CODE (lgar_py_amd/csrc/*.hpp: column physics, per-lane kernel bodies, front-capacity chain, tangent
kernels) compiled for the host by the test-only simulator (tests/devsim) against the vectors captured from the reference.
The same source runs on the GPU (tests/test_gpu_*.py repeat these checks there through the C-ABI); here logic errors
surface without a GPU.  Tolerances as on the GPU: fp64 1e-6 relative per step (observed <= 5e-8)."""
import os

import numpy as np
import pytest

from conftest import check_fault_kind, GOLDEN, golden_names


def _rel(a, b, floor=1e-6):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def _engine(g, ncol, dtype=np.float64, **kw):
    import devsim
    return devsim.SimEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=ncol,
                            dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]), ponded_depth_max=float(g["pdm"]),
                            initial_psi=float(g["initial_psi"]), wilting_point_psi=float(g["wilting_point_psi"]),
                            frozen_factor=float(g["frozen_factor"]), nint=int(g["nint"]),
                            giuh_ordinates=tuple(g["giuh_ordinates"]), dtype=dtype,
                            use_closed_form_G=bool(g["closed_form"]) if "closed_form" in g.files else False, **kw)


TRAJ = [n for n in golden_names() if not n.startswith("grad_")]


@pytest.mark.parametrize("mode", [1, 2], ids=["fast", "fast_capacity_chain"])
@pytest.mark.parametrize("name", TRAJ)
def test_device_code_fp64_trajectory_vs_reference_golden(name, mode):
    import devsim
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = _engine(g, 2, search_mode=mode)
    assert abs(float(eng.scalars[2, 0]) - float(g["init_volume"])) <= 1e-9
    f = g["forcing"][:T]
    out = eng.forward(np.repeat(f[:, 0:1], 2, 1), np.repeat(f[:, 1:2], 2, 1), series=devsim.ACC_NAMES, call_sums=True)
    for j, nm in enumerate(devsim.ACC_NAMES):
        assert _rel(out[nm][:, 0], g["acc"][:T, j]).max() <= 1e-6, nm
        assert (out[nm][:, 0] == out[nm][:, 1]).all()
    assert (eng.status == 0).all()
    nf = int(g["nfronts"][T - 1])
    fr = eng.fronts()
    assert (fr["n_fronts"] == nf).all()
    assert _rel(fr["depth"][:nf, 0], g["fronts"][T - 1, :nf, 0]).max() <= 1e-6
    assert _rel(fr["theta"][:nf, 0], g["fronts"][T - 1, :nf, 1]).max() <= 1e-6
    assert _rel(fr["psi"][:nf, 0], g["fronts"][T - 1, :nf, 2], 1e-3).max() <= 1e-5
    assert _rel(fr["k"][:nf, 0], g["fronts"][T - 1, :nf, 3], 1e-12).max() <= 1e-5
    assert _rel(fr["dzdt"][:nf, 0], g["fronts"][T - 1, :nf, 4], 1e-9).max() <= 1e-5
    assert (fr["layer"][:nf, 0] == g["front_layer"][T - 1, :nf]).all()
    assert (fr["to_bottom"][:nf, 0] == g["front_bottom"][T - 1, :nf]).all()
    for j in range(8):  # run totals (MassBalance) and the per-call sums the model surface reads
        assert _rel(float(eng.totals[j, 0]), g["acc"][:T, j].sum()) <= 1e-6
        assert _rel(float(out["call_sums"][j, 0]), g["acc"][:T, j].sum()) <= 1e-6
    if crash >= 0:  # the reference raised at this step: the column must fault here too
        f1 = g["forcing"][T:T + 1]
        eng.forward(np.repeat(f1[:, 0:1], 2, 1), np.repeat(f1[:, 1:2], 2, 1), series=())
        assert (eng.status != 0).all()
        check_fault_kind(g, eng.status)


# (search_mode, geff_mode): fast through the capacity chain, literal, mixed precision; bars on depth / theta relative to the
# reference's own value (observed: 4e-10 fast, 2e-9 literal).  The mixed-precision mode holds 1e-6 except in the step of a front
# event, where a random 1e-7 difference between consecutive Geff values is amplified ~50x (observed: one
# step of two_layer_synth1 at 7e-6, 8e-8 elsewhere): every step within 2e-5 -- the mode's bar on per-step fluxes -- and all but
# 2 % of the steps within 1e-6.
STEPWISE_MODES = {"fast": (2, 0, 1e-6, 1e-6), "literal": (0, 0, 1e-7, 1e-7), "mixed": (2, 1, 2e-5, 1e-6)}


@pytest.mark.parametrize("mode", list(STEPWISE_MODES))
@pytest.mark.parametrize("name", TRAJ)
def test_device_code_front_table_at_every_step_vs_reference_golden(name, mode):
    """north_star: "per-front depth/theta".  The engine is stepped one forcing row at a time and its WHOLE front table --
    front count, layer tags, to_bottom flags, depth, theta (layers/WettingFront.py:38-49, models/dpLGAR.py:176-298) -- is compared
    with the reference's at EVERY step, not only at the last one."""
    sm, gm, bar, usual = STEPWISE_MODES[mode]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    if mode == "literal" and T > 600:
        T = 600  # (the literal line searches take ~100x the evaluations: the head of the long fixtures)
    eng = _engine(g, 1, search_mode=sm, geff_mode=gm)
    f = g["forcing"]
    frec = g["fronts"].shape[1]
    above = 0
    for t in range(T):
        eng.forward(f[t:t + 1, 0:1], f[t:t + 1, 1:2], series=())
        nf = int(g["nfronts"][t])
        assert int(eng.n_fronts[0]) == nf, (t, int(eng.n_fronts[0]), nf)
        q = min(nf, frec)
        fl = eng.flags[:q, 0]
        assert ((fl & 0x7F) == g["front_layer"][t, :q]).all(), t
        assert ((fl >> 7) == g["front_bottom"][t, :q]).all(), t
        dz = _rel(eng.depth[:q, 0], g["fronts"][t, :q, 0]).max()
        dth = _rel(eng.theta[:q, 0], g["fronts"][t, :q, 1]).max()
        above += int(max(dz, dth) > usual)
        assert dz <= bar and dth <= bar, (t, dz, dth)
    assert above <= max(1, T // 50), (above, T)
    assert int(eng.status[0]) == 0


@pytest.mark.parametrize("name", ["synth1_phil", "phil_hourly_3000", "four_layer_synth0_600", "two_layer_phil_600",
                                  "closedG_synth1_phil", "frozen07_phil_hourly_400", "manyfronts_pulse_84", "rand06"])
def test_device_code_literal_mode_vs_reference_golden(name):
    """search_mode 0: the reference's literal line searches, update_psi pass and (fp64) operation-by-operation trapezoid."""
    import devsim
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = _engine(g, 1, search_mode=0)
    f = g["forcing"][:T]
    out = eng.forward(f[:, 0:1], f[:, 1:2], series=devsim.ACC_NAMES)
    for j, nm in enumerate(devsim.ACC_NAMES):
        assert _rel(out[nm][:, 0], g["acc"][:T, j]).max() <= 1e-7, nm  # observed <= 2e-9
    assert int(eng.n_fronts[0]) == int(g["nfronts"][T - 1]) and int(eng.status[0]) == 0


# Mixed-precision Geff (LgarDims.geff_mode = 1: fp64 state, fp32 hardware transcendentals in the trapezoid's interior nodes).
# What it reaches against the reference: a Geff value carries a RANDOM relative error of ~1e-8 (<= 2e-7),
# which the column dynamics pass on to the fluxes 1:1 except at front events, where one step's infiltration can move by up to
# ~50x that -- and a runoff that is the small difference of rainfall and infiltration moves by the same ABSOLUTE amount.  So
# the bars are: front tables 1e-6; every per-step output within 2e-5 of the water moving through the column in that step
# (i.e. of max(|value|, rainfall + ponding of the step, 1e-3 cm); observed 1.2e-6 with the host's log2f / exp2f, 1.3e-5 with
# the hardware's v_log_f32 / v_exp_f32); run totals 2e-6 of max(|total|, total rainfall); and, as a backstop, 1e-3 relative on
# every single per-step value (observed: 1.7e-4 on a 1.1e-3 cm runoff whose absolute error is 2e-7 cm).
MIXED_FLUX, MIXED_TOTAL, MIXED_STEP_BACKSTOP = 2e-5, 2e-6, 1e-3


def mixed_mode_check(acc, ref, T):
    """acc, ref: [T, NACC] per-step outputs of the mixed mode and of the reference"""
    scale = np.maximum(np.maximum(ref[:, 0:1] + ref[:, 8:9], np.abs(ref)), 1e-3)
    assert (np.abs(acc - ref) / scale).max() <= MIXED_FLUX
    assert _rel(acc, ref).max() <= MIXED_STEP_BACKSTOP
    tot, rtot = acc[:, :8].sum(0), ref[:, :8].sum(0)
    assert (np.abs(tot - rtot) / np.maximum(np.maximum(np.abs(rtot), rtot[0]), 1e-2)).max() <= MIXED_TOTAL


@pytest.mark.parametrize("name", TRAJ)
def test_device_code_mixed_precision_geff_vs_reference_golden(name):
    import devsim
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = _engine(g, 2, search_mode=2, geff_mode=1)
    f = g["forcing"][:T]
    out = eng.forward(np.repeat(f[:, 0:1], 2, 1), np.repeat(f[:, 1:2], 2, 1), series=devsim.ACC_NAMES)
    acc = np.stack([out[nm][:, 0] for nm in devsim.ACC_NAMES], 1)
    mixed_mode_check(acc, g["acc"][:T], T)
    assert all((out[nm][:, 0] == out[nm][:, 1]).all() for nm in devsim.ACC_NAMES)
    assert (eng.status == 0).all()
    nf = int(g["nfronts"][T - 1])
    fr = eng.fronts()
    assert (fr["n_fronts"] == nf).all()
    assert _rel(fr["depth"][:nf, 0], g["fronts"][T - 1, :nf, 0]).max() <= 1e-6
    assert _rel(fr["theta"][:nf, 0], g["fronts"][T - 1, :nf, 1]).max() <= 1e-6
    assert (fr["layer"][:nf, 0] == g["front_layer"][T - 1, :nf]).all()
    assert (fr["to_bottom"][:nf, 0] == g["front_bottom"][T - 1, :nf]).all()
    if crash >= 0:  # the reference raised at this step: the column must fault here too
        f1 = g["forcing"][T:T + 1]
        eng.forward(np.repeat(f1[:, 0:1], 2, 1), np.repeat(f1[:, 1:2], 2, 1), series=())
        assert (eng.status != 0).all()
        check_fault_kind(g, eng.status)


def test_mixed_precision_geff_leaf_accuracy():
    """The mixed trapezoid against the reference's literal one (library pow) over the ranges the three call sites see: no
    systematic error, <= 3e-7 relative for every input (observed on the CPU: median 8e-9, 99th percentile 7e-8, max 1.6e-7;
    the plain fp32 loop: median 1e-6, 99th percentile 6e-4)."""
    import ctypes as C
    import devsim
    L = devsim.lib(3)
    dp = np.ctypeslib.ndpointer(dtype=np.float64)
    L.devsim_geff.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, C.c_int, dp]
    rng = np.random.default_rng(0)
    n = 4000
    alpha, nn = rng.uniform(0.003, 0.009, n), rng.uniform(1.25, 1.75, n)
    ks, te, tr = np.full(n, 0.3), np.full(n, 0.46), np.full(n, 0.07)

    def run(v, t1, t2):
        o = np.zeros(n)
        L.devsim_geff(v, n, np.ascontiguousarray(t1), np.ascontiguousarray(t2), alpha, nn, ks, te, tr, 120, o)
        return o

    theta = lambda h: tr + (te - tr) * (1 + (alpha * h) ** nn) ** -(1 - 1 / nn)
    h1 = 10 ** rng.uniform(0.5, 3.3, n)
    for t1, t2 in ((theta(h1), te.copy()), (theta(h1), theta(h1 * 10 ** rng.uniform(-2, -0.05, n))),
                   (theta(h1), theta(h1 * rng.uniform(0.7, 0.98, n)))):
        ref, mix = run(2, t1, t2), run(1, t1, t2)
        e = (mix - ref) / ref
        assert np.abs(e).max() <= 3e-7 and np.median(np.abs(e)) <= 3e-8 and abs(e.mean()) <= 1e-8
        assert np.abs(run(0, t1, t2) - ref).max() <= 1e-11 * np.abs(ref).max()  # the fused fp64 trapezoid, for scale


def test_mixed_precision_flags_the_columns_the_oracle_flags():
    """Fault parity of the mixed mode on the +-10 % ensemble (the reference raises on ~13 % of it): same columns flagged,
    the others within the mixed-mode bars."""
    import devsim
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 192
    P = W.perturbed_columns(N, seed=7)
    sc = W.forcing_scale(N, seed=8)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    eng = devsim.SimEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                           ponded_depth_max=0.0, search_mode=1, geff_mode=1)
    out = eng.forward(pr, pe, series=("runoff",))
    assert ((st != 0) == (eng.status != 0)).all()
    ok = st == 0
    assert np.abs(out["runoff"][:, ok] - ro[:, ok]).max() <= MIXED_FLUX * max(1.0, np.abs(ro).max())
    scale = np.maximum(np.maximum(np.abs(acc[:8]), acc[0:1]), 1e-2)  # a column's totals against its water input
    assert (np.abs(eng.totals[:8] - acc[:8]) / scale)[:, ok].max() <= MIXED_TOTAL


def test_capacity_chain_hands_columns_over_and_resumes():
    """Columns with few fronts finish in the 8-slot kernel, the others move to 16 and 32 slots at different steps of the
    same call; results equal the single-kernel run bitwise, chunked calls included, and the per-call sums add up."""
    import devsim
    g = np.load(os.path.join(GOLDEN, "manyfronts_pulse_84.npz"))
    N, T = 6, 84
    scale = np.array([1.0, 0.0, 1.0, 0.3, 1.0, 0.6])  # column 1 never sees rain, the others grow at different rates
    pr = g["forcing"][:T, 0:1] * scale[None, :]
    pe = np.zeros_like(pr)
    a = _engine(g, N, search_mode=1)
    ref = a.forward(pr, pe, series=devsim.ACC_NAMES, basin=("runoff", "infiltration"), call_sums=True)
    assert a.n_fronts.max() > 16 and a.n_fronts.min() == 3 and (a.status == 0).all()
    b = _engine(g, N, search_mode=2)
    got = b.forward(pr, pe, series=devsim.ACC_NAMES, basin=("runoff", "infiltration"), call_sums=True)
    for nm in ref:
        if nm.startswith("basin") or nm == "call_sums":  # sums are split at the hand-over points: last-bit differences
            assert np.allclose(got[nm], ref[nm], rtol=1e-13, atol=1e-15), nm
        else:
            assert np.array_equal(got[nm], ref[nm]), nm
    for x, y in ((a.depth, b.depth), (a.theta, b.theta), (a.psi, b.psi), (a.dzdt, b.dzdt), (a.flags, b.flags),
                 (a.n_fronts, b.n_fronts), (a.scalars, b.scalars), (a.status, b.status)):
        assert np.array_equal(x, y)
    assert np.allclose(a.totals, b.totals, rtol=1e-14, atol=0)
    c = _engine(g, N, search_mode=2)  # ragged chunks: hand-overs happen in different calls
    parts = []
    for lo, hi in ((0, 7), (7, 30), (30, 31), (31, 70), (70, 84)):
        parts.append(c.forward(pr[lo:hi], pe[lo:hi], series=("runoff", "ending_volume"), call_sums=True))
        assert (c.status == 0).all()  # LGAR_ST_RESUME never survives a call
    assert np.array_equal(np.concatenate([p["runoff"] for p in parts]), ref["runoff"])
    assert np.array_equal(np.concatenate([p["ending_volume"] for p in parts]), ref["ending_volume"])
    assert np.allclose(sum(p["call_sums"][:8] for p in parts), ref["call_sums"][:8], rtol=1e-13, atol=1e-15)
    assert np.array_equal(c.n_fronts, a.n_fronts) and np.array_equal(c.depth, a.depth)


def test_front_overflow_is_flagged_at_the_reference_state_limit():
    """The reference's lists are unbounded; here a column holds at most front_slots (<= 32) fronts: one more -> the
    column stops with LGAR_ST_OVERFLOW, in every mode, with the state arrays never written past their rows."""
    import devsim
    g = np.load(os.path.join(GOLDEN, "manyfronts_pulse_84.npz"))
    f = np.concatenate([g["forcing"], g["forcing"][:40] * 0 + np.array([[0.02, 0.0]])])  # keep pulsing past 32 fronts
    T = g["forcing"].shape[0]
    for mode in (0, 1, 2):
        for slots in (None, 10):
            eng = _engine(g, 1, search_mode=mode, front_slots=slots)
            pr = np.concatenate([g["forcing"][:, 0], np.tile([0.02, 0.0], 30)])[:, None]
            eng.forward(pr, np.zeros_like(pr), series=())
            lim = slots or 32
            assert int(eng.status[0]) & 8, (mode, slots)
            assert int(eng.n_fronts[0]) == lim
            assert eng.depth.shape[0] == lim


def test_device_code_fp32_close_to_reference():
    """The fp32 instantiation (bench.py's configuration; host libm stands in for v_log_f32 / v_exp_f32) on the golden
    cases of its workload family: run totals within 5e-3 of the reference."""
    import devsim
    for name in ("synth1_phil", "synth1_pert0", "synth1_pert3", "synth2_phil", "phil_pert1_500"):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        eng = _engine(g, 1, dtype=np.float32)
        f = g["forcing"]
        eng.forward(f[:, 0:1], f[:, 1:2], series=())
        assert int(eng.status[0]) == 0
        ref = g["acc"][:, :8].sum(0)
        got = eng.totals[:8, 0].astype(np.float64)
        scale = max(ref[0], 1.0)  # precipitation scale
        assert np.abs(got - ref).max() <= 5e-3 * scale, (name, got, ref)
        assert abs(float(eng.totals[9, 0]) - g["acc"][-1, 9]) <= 5e-3 * g["acc"][-1, 9]


GRADS = [n for n in golden_names() if n.startswith("grad_")]


@pytest.mark.parametrize("mode", [1, 2, 0], ids=["fast", "fast_capacity_chain", "literal"])
@pytest.mark.parametrize("name", GRADS)
def test_device_tangent_matches_reference_autograd(name, mode):
    """Forward-mode tangents of the device code (Dual numbers) contracted with d loss / d runoff_t reproduce the gradients
    torch autograd produced through the reference (loss = mean(runoff^2)): nominal, 4-layer, +-10 % ensemble members,
    wide-range ensemble members, and a column with up to 19 fronts."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    f = g["forcing"]
    T = f.shape[0]
    L = len(g["alpha"])
    eng = _engine(g, 1, search_mode=mode)
    out = eng.forward(f[:, 0:1], f[:, 1:2], series=("runoff",))
    r = out["runoff"][:, 0]
    loss = float(np.mean(r * r))
    assert abs(loss - float(g["loss"])) <= 1e-9 * float(g["loss"])
    w = (2.0 * r / T)[:, None]
    for kind, ref in (("alpha", g["d_alpha"]), ("n", g["d_n"]), ("ksat", g["d_ksat"])):
        ref = np.nan_to_num(ref, nan=0.0)  # None in the reference = no dependence
        if kind == "ksat":  # the reference's Parameter is Ksat x frozen_factor (models/dpLGAR.py:57), the engine's input is Ksat
            ref = ref * float(g["frozen_factor"])
        got = np.zeros(L)
        for l in range(L):
            d = np.zeros((L, 1))
            d[l] = 1.0
            gr, _, st = eng.tangent({kind: d}, f[:, 0:1], f[:, 1:2], w_runoff=w)
            assert int(st[0]) == 0
            got[l] = gr[0]
        assert np.abs(got - ref).max() <= 1e-6 * np.abs(ref).max(), (kind, got, ref)


def test_tangent_capacity_chain_and_status():
    """Tangent kernels: the 8-slot kernel hands the many-front column to the 32-slot kernel (same gradient as the
    single-kernel run), and a column that overflows even that reports it in the tangent status."""
    g = np.load(os.path.join(GOLDEN, "grad_manyfronts_60.npz"))
    f = g["forcing"]
    T = f.shape[0]
    N = 3
    scale = np.array([1.0, 0.0, 0.5])
    pr, pe = f[:, 0:1] * scale[None, :], np.zeros((T, N))
    w = np.ones((T, N))
    d = np.zeros((3, N))
    d[0] = 1.0
    g1, _, s1 = _engine(g, N, search_mode=1).tangent({"ksat": d}, pr, pe, w_runoff=w)
    g2, _, s2 = _engine(g, N, search_mode=2).tangent({"ksat": d}, pr, pe, w_runoff=w)
    assert (s1 == 0).all() and (s2 == 0).all()
    assert np.array_equal(g1, g2) and g1[0] != 0.0 and g1[1] == 0.0
    long = np.concatenate([np.load(os.path.join(GOLDEN, "manyfronts_pulse_84.npz"))["forcing"][:, 0], np.tile([0.02, 0.0], 30)])[:, None]
    gl, _, sl = _engine(g, 1, search_mode=2).tangent({"ksat": d[:, :1]}, long, np.zeros_like(long), w_runoff=np.ones_like(long))
    assert int(sl[0]) & 8


def test_columns_outside_the_reference_domain_are_flagged_like_the_oracle():
    """+-10 % perturbed columns under the synth_1 storm: ~13 % of them make the reference raise ValueError (negative pow
    base in insert_water's Geff, quirk q3).  Device code (every mode) and oracle must flag exactly the same columns and
    agree on all the others."""
    import devsim
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 192
    P = W.perturbed_columns(N, seed=7)
    sc = W.forcing_scale(N, seed=8)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    assert 0 < (st != 0).mean() < 0.5
    for mode in (0, 1, 2):
        eng = devsim.SimEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                               ponded_depth_max=0.0, search_mode=mode)
        out = eng.forward(pr, pe, series=("runoff",))
        assert ((st != 0) == (eng.status != 0)).all(), mode
        ok = st == 0
        assert np.abs(out["runoff"][:, ok] - ro[:, ok]).max() <= 1e-6 * max(1.0, np.abs(ro).max()), mode
        assert _rel(eng.totals[:8][:, ok], acc[:8][:, ok], 1e-3).max() <= 1e-6, mode


def test_forcing_broadcast_equals_replicated_forcing():
    """LgarDims.forcing_columns: soil column c reads forcing column c % Nf.  One basin series for every column (Nf = 1, what
    the reference's Data yields) and the direction-major layout of the differentiable path (Nf = N / D) give bitwise the
    results of explicitly replicated forcing, forward and tangent."""
    import devsim
    g = np.load(os.path.join(GOLDEN, "synth0_phil_1500.npz"))
    f = g["forcing"][:200]
    N = 6
    a = _engine(g, N)
    full = a.forward(np.repeat(f[:, 0:1], N, 1), np.repeat(f[:, 1:2], N, 1), series=("runoff", "AET"))
    b = _engine(g, N)
    one = b.forward(f[:, 0:1], f[:, 1:2], series=("runoff", "AET"))
    for nm in full:
        assert np.array_equal(full[nm], one[nm])
    assert np.array_equal(a.theta, b.theta) and np.array_equal(a.totals, b.totals)
    sc = np.array([1.0, 0.5, 0.8])
    pr3, pe3 = f[:, 0:1] * sc[None, :], f[:, 1:2] * np.ones((1, 3))
    c = _engine(g, N)
    half = c.forward(pr3, pe3, series=("runoff",))  # columns 0..2 and 3..5 see forcing columns 0..2
    d = _engine(g, N)
    rep = d.forward(np.tile(pr3, (1, 2)), np.tile(pe3, (1, 2)), series=("runoff",))
    assert np.array_equal(half["runoff"], rep["runoff"])
    w = np.linspace(0.5, 1.5, 200)[:, None] * np.ones((1, 3))
    dirs = np.zeros((3, N))
    dirs[0, :3] = 1.0   # first direction: layer-0 Ksat on columns 0..2
    dirs[1, 3:] = 1.0   # second direction: layer-1 Ksat on columns 3..5 (same soils, same forcing)
    g1, _, s1 = c.tangent({"ksat": dirs}, pr3, pe3, w_runoff=w)
    g2, _, s2 = d.tangent({"ksat": dirs}, np.tile(pr3, (1, 2)), np.tile(pe3, (1, 2)), w_runoff=np.tile(w, (1, 2)))
    assert np.array_equal(g1, g2) and (s1 == 0).all() and np.abs(g1).max() > 0
    # forcing_group: G consecutive columns share a forcing column (the differentiable path's interleaved directions)
    e = _engine(g, N)
    grp = e.forward(pr3, pe3, series=("runoff",), forcing_group=2)  # columns (0,1), (2,3), (4,5) see forcing columns 0, 1, 2
    assert np.array_equal(grp["runoff"], rep["runoff"][:, [0, 3, 1, 4, 2, 5]])
    dirs2 = np.zeros((3, N))
    dirs2[0, 0::2] = 1.0
    dirs2[1, 1::2] = 1.0
    g3, _, s3 = e.tangent({"ksat": dirs2}, pr3, pe3, w_runoff=w, forcing_group=2)
    assert np.array_equal(g3, g2[[0, 3, 1, 4, 2, 5]]) and (s3 == 0).all()


def test_nan_in_the_forcing_is_flagged():
    """The reference lets a NaN forcing value through (comparisons with it are false, the step's runoff is NaN, nothing
    raises); the engine flags the column (LGAR_ST_NAN) so that the caller hears about the bad datum."""
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    f = g["forcing"].copy()
    f[100, 0] = np.nan
    for mode in (0, 1):
        e = _engine(g, 2, search_mode=mode)
        pr = np.stack([f[:, 0], g["forcing"][:, 0]], 1)
        pe = np.stack([f[:, 1], g["forcing"][:, 1]], 1)
        e.forward(pr, pe, series=("runoff",))
        assert e.status[0] & 1 and e.status[1] == 0


@pytest.mark.parametrize("name", ["rand03", "synth2_phil", "bench_col10707", "five_layer_synth1", "manyfronts_pulse_84"])
def test_giuh_queue_in_place_survives_cut_launches_and_matches_the_reference_queue(name):
    """The mixed-precision kernels update the GIUH queue in place in the `scalars` rows of the state (Column::GIUH_MEM) instead of
    carrying it in registers; the flag "something is queued" is rebuilt from those rows at every launch.  A run cut into four
    launches must equal the one-launch run bit for bit -- series, scalars, front table --, and at every cut the queue rows must
    hold the reference's own queue (lgar/giuh.py:8-20; fixtures: `giuh_queue[t]`): to 1e-9 in the native mode, to the
    mixed-mode flux bar otherwise."""
    import devsim
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    T = g["forcing"].shape[0]
    f = g["forcing"]
    pr, pe = np.repeat(f[:, 0:1], 2, 1), np.repeat(f[:, 1:2], 2, 1)
    ng = len(g["giuh_ordinates"])
    cuts = (7, 31, 32, T)
    for geff_mode, bar in ((1, MIXED_FLUX), (0, 1e-9)):
        one = _engine(g, 2, search_mode=2, geff_mode=geff_mode)
        whole = one.forward(pr, pe, series=devsim.ACC_NAMES)
        cut = _engine(g, 2, search_mode=2, geff_mode=geff_mode)
        parts, t0 = [], 0
        scale = max(float(np.abs(g["giuh_queue"]).max()), 1e-6)
        for t1 in cuts:
            if t1 <= t0:
                continue
            parts.append(cut.forward(pr[t0:t1], pe[t0:t1], series=devsim.ACC_NAMES))
            q = cut.scalars[3:3 + ng, 0]
            assert np.abs(q - g["giuh_queue"][t1 - 1]).max() <= bar * scale, (geff_mode, t1)
            assert (cut.scalars[3 + ng:, 0] == 0).all()
            t0 = t1
        for nm in devsim.ACC_NAMES:
            assert np.array_equal(np.concatenate([p[nm] for p in parts], 0), whole[nm]), (geff_mode, nm)
        assert np.array_equal(cut.scalars, one.scalars) and np.array_equal(cut.status, one.status)
        a, b = cut.fronts(), one.fronts()
        assert all(np.array_equal(a[k], b[k]) for k in a)
        assert (whole["giuh_runoff"][:, 0] > 0).sum() > 3  # the fixture routes runoff
