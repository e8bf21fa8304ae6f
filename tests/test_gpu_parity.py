"""GPU (-m gpu): the HIP path, called through the C-ABI, against the golden vectors captured from the
reference and against the CPU oracle on the same seeded inputs.

Tolerances: fp64 1e-6 relative per step on every accumulator (north_star; observed <= 1e-9);
fp32 (the throughput configuration) 5e-3 relative on run totals (99th percentile of columns) -- fp32 cannot hold the reference's
absolute 1e-12 mass tolerance, see DESIGN.md."""
import os

import numpy as np
import pytest

from conftest import check_fault_kind, GOLDEN, golden_names

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _rel(a, b, floor=1e-6):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def _engine(g, ncol, dtype, **kw):
    import lgar_py_amd as lg
    return lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=ncol,
                         dt_h=float(g["dt_h"]), num_subcycles=int(g["num_subcycles"]),
                         ponded_depth_max=float(g["pdm"]), initial_psi=float(g["initial_psi"]),
                         wilting_point_psi=float(g["wilting_point_psi"]), frozen_factor=float(g["frozen_factor"]),
                         nint=int(g["nint"]), giuh_ordinates=tuple(g["giuh_ordinates"]), dtype=dtype,
                         use_closed_form_G=bool(g["closed_form"]) if "closed_form" in g.files else False, **kw)


def _forcing(g, ncol, sl=slice(None)):
    f = torch.tensor(g["forcing"][sl])
    T = f.shape[0]
    return f[:, 0:1].expand(T, ncol).contiguous(), f[:, 1:2].expand(T, ncol).contiguous()


TRAJ = [n for n in golden_names() if not n.startswith("grad_")]


@pytest.mark.parametrize("mode", [1, 2, 0], ids=["fast_search", "fast_capacity_chain", "literal_search"])
@pytest.mark.parametrize("name", TRAJ)
def test_fp64_trajectory_vs_reference_golden(name, mode):
    """All search modes against the reference: 0 = its literal fixed-step line searches (and trapezoid), 1 (the default,
    what bench.py measures) = Newton / closed-form-jump searches to the same tolerances, 2 = 1 with the front-capacity
    chain (8 -> 16 -> 32 slots) forced for this small job."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ncol = 67  # one full wave + a ragged tail
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = _engine(g, ncol, torch.float64, search_mode=mode)
    assert abs(float(eng.ending_volume[0]) - float(g["init_volume"])) <= 1e-9
    pr, pe = _forcing(g, ncol, slice(0, T))
    out = eng.forward(pr, pe, series=lg.ACC_NAMES)
    for j, nm in enumerate(lg.ACC_NAMES):
        got = out[nm].cpu().numpy()
        assert _rel(got[:, 0], g["acc"][:T, j]).max() <= 1e-6, nm
        assert (got == got[:, :1]).all(), "replicated columns must be bit-identical"
    fr = eng.fronts()
    nf = int(g["nfronts"][T - 1])
    assert (fr["n_fronts"] == nf).all()
    assert _rel(fr["depth"][:nf, 0], g["fronts"][T - 1, :nf, 0]).max() <= 1e-6
    assert _rel(fr["theta"][:nf, 0], g["fronts"][T - 1, :nf, 1]).max() <= 1e-6
    assert _rel(fr["psi"][:nf, 0], g["fronts"][T - 1, :nf, 2], 1e-3).max() <= 1e-5
    assert _rel(fr["k"][:nf, 0], g["fronts"][T - 1, :nf, 3], 1e-12).max() <= 1e-5   # K(theta), deepest front keeps its initial K
    assert _rel(fr["dzdt"][:nf, 0], g["fronts"][T - 1, :nf, 4], 1e-9).max() <= 1e-5
    assert (fr["layer"][:nf, 0] == g["front_layer"][T - 1, :nf]).all()
    assert (fr["to_bottom"][:nf, 0] == g["front_bottom"][T - 1, :nf]).all()
    # run totals (what MassBalance accumulates)
    for j in range(8):
        assert _rel(float(eng.totals[j, 0]), g["acc"][:T, j].sum()) <= 1e-6
    if crash >= 0:
        # the reference raised at this step (ValueError / AttributeError): every column must fault here too
        pr1, pe1 = _forcing(g, ncol, slice(T, T + 1))
        with pytest.raises(lg.LgarStatusError):
            eng.forward(pr1, pe1)
        assert bool((eng.status != 0).all())
        check_fault_kind(g, eng.status.cpu().numpy())


# engine settings per mode and the bar on depth / theta (relative to the reference's own value); "fast" lets the library choose
# its kernel for a job this small (cooperating lanes), "fast_capacity_chain" forces the one-lane-per-column kernels of the big jobs
# The mixed-precision mode holds 1e-6 except in the step of a front event, where a random 1e-7 difference between consecutive
# Geff values is amplified ~50x: every step within 2e-5 -- the mode's bar on per-step fluxes -- and all but
# 2 % of the steps within 2e-6.
STEPWISE_MODES = {"fast": (dict(search_mode=1), 1e-6, 1e-6), "fast_capacity_chain": (dict(search_mode=2), 1e-6, 1e-6),
                  "literal": (dict(search_mode=0), 1e-7, 1e-7), "mixed": (dict(search_mode=2, geff_precision="f32"), 2e-5, 2e-6),
                  "mixed_cooperating_lanes": (dict(search_mode=1, geff_precision="f32"), 2e-5, 2e-6)}


@pytest.mark.parametrize("mode", list(STEPWISE_MODES))
@pytest.mark.parametrize("name", TRAJ)
def test_fp64_front_table_at_every_step_vs_reference_golden(name, mode):
    """north_star: "per-front depth/theta".  The HIP engine is stepped one forcing row at a time through the C-ABI and its WHOLE
    front table -- front count, layer tags, to_bottom flags, depth, theta (layers/WettingFront.py:38-49,
    models/dpLGAR.py:176-298) -- is compared with the reference's at EVERY step, not only at the last one."""
    kw, bar, usual = STEPWISE_MODES[mode]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    if mode == "literal" and T > 600:
        T = 600  # (the literal line searches take ~100x the evaluations: the head of the long fixtures)
    ncol = 2
    eng = _engine(g, ncol, torch.float64, **kw)
    pr, pe = _forcing(g, ncol, slice(0, T))
    F = eng.depth.shape[0]
    dev = eng.depth.device
    Z = torch.empty(T, F, dtype=torch.float64, device=dev)
    TH = torch.empty(T, F, dtype=torch.float64, device=dev)
    FL = torch.empty(T, F, dtype=torch.uint8, device=dev)
    NF = torch.empty(T, dtype=torch.int32, device=dev)
    for t in range(T):
        eng.forward(pr[t:t + 1], pe[t:t + 1], series=(), check=False)
        Z[t], TH[t], FL[t], NF[t] = eng.depth[:, 0], eng.theta[:, 0], eng.flags[:, 0], eng.n_fronts[0]
    assert bool((eng.status == 0).all())
    assert torch.equal(eng.depth[:, 0], eng.depth[:, 1]) and torch.equal(eng.theta[:, 0], eng.theta[:, 1])
    Z, TH, FL, NF = Z.cpu().numpy(), TH.cpu().numpy(), FL.cpu().numpy(), NF.cpu().numpy()
    assert (NF == g["nfronts"][:T]).all(), int(np.argmax(NF != g["nfronts"][:T]))
    frec = min(g["fronts"].shape[1], F)
    live = np.arange(frec)[None, :] < np.minimum(NF, frec)[:, None]
    assert ((FL[:, :frec] & 0x7F)[live] == g["front_layer"][:T, :frec][live]).all()
    assert ((FL[:, :frec] >> 7)[live] == g["front_bottom"][:T, :frec][live]).all()
    err = np.maximum(_rel(Z[:, :frec], g["fronts"][:T, :frec, 0]), _rel(TH[:, :frec], g["fronts"][:T, :frec, 1]))
    err = np.where(live, err, 0.0).max(axis=1)  # worst front of every step
    assert err.max() <= bar, (int(err.argmax()), float(err.max()))
    assert int((err > usual).sum()) <= max(1, T // 50), (int((err > usual).sum()), T, float(err.max()))


def test_config2_10k_replicated_phillipsburg_fp64():
    """BASELINE configs[1]: 10k replicated Phillipsburg columns, fp64, every column == reference to 1e-6 rel."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    N = 10_000
    eng = _engine(g, N, torch.float64)
    pr, pe = _forcing(g, N)
    out = eng.forward(pr, pe, series=("infiltration", "runoff", "AET", "percolation", "ending_volume"))
    for nm in out:
        got = out[nm]
        ref = torch.tensor(g["acc"][:, lg.ACC_NAMES.index(nm)], device=got.device)[:, None]
        rel = ((got - ref).abs() / ref.abs().clamp_min(1e-6)).max().item()
        assert rel <= 1e-6, (nm, rel)
    cum = out["infiltration"].sum(0)
    assert abs(float(cum[0]) - 19.786600000000007) <= 1e-6 * 19.8
    assert float((cum - cum[0]).abs().max()) == 0.0
    fr = eng.fronts()
    nf = int(g["nfronts"][-1])
    assert (fr["n_fronts"] == nf).all()
    assert _rel(fr["depth"][:nf], g["fronts"][-1, :nf, 0:1]).max() <= 1e-6
    assert _rel(fr["theta"][:nf], g["fronts"][-1, :nf, 1:2]).max() <= 1e-6


def test_chunked_run_equals_single_run():
    """State persistence across calls: T steps in one launch == the same T steps in ragged chunks (bitwise)."""
    g = np.load(os.path.join(GOLDEN, "synth0_phil_1500.npz"))
    ncol = 3
    pr, pe = _forcing(g, ncol, slice(0, 300))
    a = _engine(g, ncol, torch.float64)
    full = a.forward(pr, pe, series=("runoff", "AET", "ending_volume"))
    b = _engine(g, ncol, torch.float64)
    parts = {k: [] for k in full}
    for lo, hi in ((0, 1), (1, 8), (8, 150), (150, 299), (299, 300)):
        o = b.forward(pr[lo:hi], pe[lo:hi], series=tuple(full))
        for k in full:
            parts[k].append(o[k])
    for k in full:
        assert torch.equal(torch.cat(parts[k]), full[k]), k
    # run totals are accumulated per call and then added: equal up to the last bits; the state itself is bitwise equal
    assert torch.allclose(a.totals, b.totals, rtol=1e-13, atol=1e-15) and torch.equal(a.depth, b.depth) and torch.equal(a.theta, b.theta)
    # reset() really is set_internal_states: re-running reproduces the first run bitwise
    a.reset()
    again = a.forward(pr, pe, series=("runoff",))
    assert torch.equal(again["runoff"], full["runoff"])


def test_heterogeneous_columns_vs_oracle_fp64():
    """Seeded +-10 % perturbed columns with per-column forcing scale (the bench workload's shape at a size the
    oracle finishes in seconds): every column within 1e-6 rel of the oracle; mass closes per column."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 512
    P = W.perturbed_columns(N, seed=7)
    sc = W.forcing_scale(N, seed=8)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    for mode in (0, 1, 2):  # literal searches; the default fast mode bench.py times; fast with the capacity chain
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=torch.float64, search_mode=mode)
        v0 = eng.ending_volume.clone()
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff", "percolation"), check=False)
        # Some perturbed columns leave the reference's domain of validity (it raises ValueError: negative pow base in
        # insert_water's Geff, quirk q3); oracle and kernel must flag exactly the same columns, and still agree on them.
        gst = eng.status.cpu().numpy()
        assert ((st != 0) == (gst != 0)).all(), mode
        assert 0 < (st != 0).mean() < 0.5
        with pytest.raises(lg.LgarStatusError):
            eng.check_status()
        got = out["runoff"].cpu().numpy()
        assert np.abs(got - ro).max() <= 1e-6 * max(1.0, np.abs(ro).max()), mode
        tot = eng.totals.cpu().numpy()
        assert _rel(tot[:8], acc[:8], 1e-3).max() <= 1e-6, mode
        assert _rel(tot[9], acc[9]).max() <= 1e-9
        # size-independent property: the global mass balance of MassBalance.report_mass (MassBalance.py:84-92) closes
        # for the bulk of the columns.  It is NOT an invariant of the reference's algorithm (its own synth3 run with
        # ponding leaves 0.108 cm unaccounted, tests/golden/synth3_generic.npz), so only the median is asserted here;
        # column-by-column the totals above already equal the oracle's.
        err = v0.cpu().numpy() + tot[0] - tot[4] - tot[2] - tot[8] - tot[5] - tot[9]
        assert np.median(np.abs(err[st == 0])) <= 1e-8


def test_fp32_throughput_configuration_vs_oracle():
    """fp32 (BASELINE configs[2]) against the fp64 oracle on the same seeded columns: run totals within 5e-3
    relative (99th percentile; 5e-2 worst column), basin runoff within 1e-3, no faulted column."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 1024
    P = W.perturbed_columns(N, seed=0)
    sc = W.forcing_scale(N, seed=1)
    f = W.synth1_forcing()
    pr = f[:, 0:1] * sc[None, :]
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                        ponded_depth_max=0.0, dtype=torch.float32)
    out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff",), check=False)
    gst = eng.status.cpu().numpy()
    # Which columns leave the reference's domain of validity hinges on a psi tie at the 1e-8 level (free-drainage
    # front choice right after a layer crossing), so the fp32 and fp64 sets differ; compare where both are valid.
    ok = (st == 0) & (gst == 0)
    assert ok.mean() > 0.5 and (gst != 0).mean() < 0.2
    tot = eng.totals.double().cpu().numpy()
    for j in (0, 3, 9):  # precip, infiltration, ending volume: O(1..50) cm quantities
        r = _rel(tot[j][ok], acc[j][ok], 1e-2)
        assert np.percentile(r, 99) <= 5e-3 and r.max() <= 5e-2, (j, r.max())
    dro = np.abs(tot[4] - acc[4])[ok] / np.maximum(acc[0][ok], 1.0)  # runoff error against the precipitation scale
    assert np.percentile(dro, 99) <= 5e-3 and dro.max() <= 5e-2, dro.max()
    basin = out["runoff"].double().cpu().numpy()[:, ok].sum()
    assert abs(basin - ro[:, ok].sum()) <= 1e-3 * abs(ro[:, ok].sum())
    err = tot[0] - tot[3] - tot[4] - tot[8]  # precip = infiltration + runoff + ponded
    assert np.abs(err[ok]).max() <= 1e-4


def test_leaf_kats_on_gpu():
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "leaf_kats.npz"))
    soils = g["soils"]
    S = len(soils)

    def bc(arr_len):
        cols = [np.repeat(soils[:, j], arr_len) for j in range(5)]
        return dict(alpha=cols[0], n=cols[1], ksat=cols[2], theta_e=cols[3], theta_r=cols[4])

    H, E = len(g["hs"]), len(g["ses"])
    hs = np.tile(g["hs"], S)
    ses = np.tile(g["ses"], S)
    for op, x, want, tol in (("theta_from_h", hs, g["theta_from_h"], 1e-13), ("se_from_h", hs, g["se_from_h"], 1e-13),
                             ("k_from_se", ses, g["k_from_se"], 1e-10), ("h_from_se", ses, g["h_from_se"], 1e-10)):
        got = lg.leaf_batch(op, x, **bc(len(x) // S)).cpu().numpy().reshape(want.shape)
        assert _rel(got, want, 1e-300).max() <= tol, op
    t1, t2 = g["geff_theta1"].ravel(), g["geff_theta2"].ravel()
    got = lg.leaf_batch("geff", t1, t2, **bc(g["geff"].shape[1])).cpu().numpy().reshape(g["geff"].shape)
    # fused-node trapezoid (four lean ~2-ulp log2/exp2 per node, re-associated formula, no 1e-12 nudge on interior nodes):
    # the extreme pairs (Se 0.02 -> 1, integral dominated by the last node at h ~ 1e-5 cm) are the least well conditioned
    assert _rel(got, g["geff"], 1e-300).max() <= 1e-8
    # the literal trapezoid (what search_mode 0 runs in fp64): the reference's own operations, node by node
    lit = lg.leaf_batch("geff_literal", t1, t2, **bc(g["geff"].shape[1])).cpu().numpy().reshape(g["geff"].shape)
    assert _rel(lit, g["geff"], 1e-300).max() <= 1e-10
    # fp32 fast-pow path: a few ulp of fp32 on the trapezoid
    # (inputs clamped to theta_e after the cast: a theta rounded above theta_e has Se > 1, which is outside the domain)
    te32 = bc(g["geff"].shape[1])["theta_e"].astype(np.float32)
    t1c, t2c = np.minimum(t1.astype(np.float32), te32), np.minimum(t2.astype(np.float32), te32)
    got32 = lg.leaf_batch("geff", t1c, t2c, dtype=torch.float32, **bc(g["geff"].shape[1])).cpu().numpy().reshape(g["geff"].shape)
    assert _rel(got32, g["geff"], 1e-3).max() <= 5e-3
    A = g["aet"]
    psis, pets, dts = g["aet_psis"], g["aet_pets"], g["aet_dts"]
    for di, dt in enumerate(dts):
        x = np.tile(np.repeat(psis, len(pets)), S)
        y = np.tile(np.tile(pets, len(psis)), S)
        got = lg.leaf_batch("aet", x, y, float(dt), **bc(len(psis) * len(pets))).cpu().numpy().reshape(S, len(psis), len(pets))
        assert np.abs(got - A[:, :, :, di]).max() <= 1e-12


def test_domain_bottom_raises_like_reference():
    """A front reaching the domain bottom kills the reference (AttributeError, Layer.py:980); here the column's
    status gets LGAR_ST_BOTTOM and the wrapper raises a ValueError subclass."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    eng = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], [4.0, 4.0, 4.0], n_columns=5,
                        dt_h=float(g["dt_h"]), ponded_depth_max=0.0)
    pr, pe = _forcing(g, 5)
    with pytest.raises(lg.LgarStatusError, match="domain bottom"):
        eng.forward(pr, pe)
    assert isinstance(lg.LgarStatusError("x"), ValueError)
    assert int(eng.status[0]) & 32


def test_bad_arguments_are_rejected():
    import ctypes as C
    import lgar_py_amd as lg
    from lgar_py_amd import _capi
    with pytest.raises(lg.LgarError):
        lg.LgarEngine([1e-2] * 7, [1.5] * 7, [1.0] * 7, [0.4] * 7, [0.1] * 7, [10.0] * 7, n_columns=4)  # 7 layers: not compiled in
    with pytest.raises(lg.LgarError, match="n > 1"):
        lg.LgarEngine([1e-2] * 3, [1.0, 1.5, 1.5], [1.0] * 3, [0.4] * 3, [0.1] * 3, [10.0] * 3, n_columns=4)
    with pytest.raises(lg.LgarError, match="theta_e > theta_r"):
        lg.LgarEngine([1e-2] * 3, [1.5] * 3, [1.0] * 3, [0.4, 0.05, 0.4], [0.1] * 3, [10.0] * 3, n_columns=4)
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    eng = _engine(g, 4, torch.float64)
    with pytest.raises(lg.LgarError):
        eng.forward(torch.zeros(3, 5), torch.zeros(3, 5))  # wrong column count
    lib = _capi.load()
    assert lib.lgar_forward(None, None, None, None, None, None, 1, None) == -1
    d = _capi.LgarDims()
    C.memmove(C.byref(d), C.byref(eng.dims), C.sizeof(d))
    d.n_layers = 9
    assert lib.lgar_state_init(C.byref(d), C.byref(eng._params), C.byref(eng._state), eng.status.data_ptr(), 1, None) == -1
    # a state the library did not produce (zero fronts) is rejected per column, not integrated
    eng.n_fronts.zero_()
    with pytest.raises(lg.LgarStatusError, match="structural"):
        eng.forward(torch.zeros(2, 4), torch.zeros(2, 4))
    eng.reset()
    # empty run is a no-op
    out = eng.forward(torch.zeros(0, 4), torch.zeros(0, 4))
    assert out["runoff"].shape == (0, 4)


def test_nan_in_the_forcing_is_flagged_on_gpu():
    """A NaN forcing value slips through the reference silently; the engine flags that column and only that column."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    for dtype in (torch.float64, torch.float32):
        eng = _engine(g, 130, dtype)
        pr, pe = _forcing(g, 130)
        pr = pr.clone()
        pr[100, 77] = float("nan")
        eng.forward(pr, pe, series=(), check=False)
        st = eng.status.cpu().numpy()
        assert st[77] & 1 and (np.delete(st, 77) == 0).all()
        with pytest.raises(lg.LgarStatusError):
            eng.check_status()


def test_heterogeneous_hourly_with_pet_vs_oracle_fp64():
    """Perturbed columns under the hourly Phillipsburg forcing (rain + PET: AET, dry-over-wet, merges, base case)
    scaled per column: kernel (literal searches) vs oracle column by column, 600 steps."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    N, T = 256, 600
    P = W.perturbed_columns(N, seed=21)
    sc = W.forcing_scale(N, 0.5, 2.0, seed=22)
    pr = g["forcing"][:T, 0:1] * sc[None, :]
    pe = g["forcing"][:T, 1:2] * np.ones((1, N))
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=2.0, dt_h=1.0)
    for mode, tol in ((0, 1e-6), (1, 1e-6)):
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=1.0,
                            ponded_depth_max=2.0, dtype=torch.float64, search_mode=mode)
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff", "AET", "infiltration"), check=False)
        gst = eng.status.cpu().numpy()
        if mode == 0:
            assert ((st != 0) == (gst != 0)).all()
        ok = (st == 0) & (gst == 0)
        assert ok.mean() > 0.9
        tot = eng.totals.cpu().numpy()
        assert _rel(tot[:8, ok], acc[:8, ok], 1e-3).max() <= tol, mode
        assert _rel(tot[9, ok], acc[9, ok]).max() <= tol
        assert tot[2, ok].min() > 0  # AET path exercised
        fr = eng.fronts()
        assert fr["n_fronts"][ok].max() <= 12


def test_wet_hourly_ensemble_faults_where_the_oracle_does():
    """4096 perturbed columns under the hourly forcing with rain multipliers up to 3 (tools/parity_sweep.py found this
    shape): a few columns hit the reference's ValueError inside fix_dry_over_wet (psi of another layer's theta, Se > 1,
    Layer.py:1117-1143; fixture crash_dry_over_wet_300 is the reference's own run of one of them) -- a NaN that update_psi
    overwrites before anything reads it.  Every mode must flag exactly the oracle's columns and agree to 1e-6 elsewhere."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    N, T = 4096, 600
    P = W.perturbed_columns(N, seed=400)
    sc = W.forcing_scale(N, 0.5, 3.0, seed=500)
    pr = g["forcing"][:T, 0:1] * sc[None, :]
    pe = g["forcing"][:T, 1:2] * np.ones((1, N))
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe,
                                    pdm=2.0, dt_h=1.0)
    assert st[1270] != 0 and 0 < (st != 0).sum() < 100
    for mode in (0, 1, 2):
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=1.0,
                            ponded_depth_max=2.0, dtype=torch.float64, search_mode=mode)
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff",), check=False)
        gst = eng.status.cpu().numpy()
        assert ((st != 0) == (gst != 0)).all(), mode
        ok = st == 0
        assert np.abs(out["runoff"].cpu().numpy() - ro)[:, ok].max() <= 1e-6 * max(1.0, np.abs(ro).max()), mode
        tot = eng.totals.cpu().numpy()
        assert _rel(tot[:8, ok], acc[:8, ok], 1e-3).max() <= 1e-6, mode


# columns of W.ensemble_columns(512, seed=3) on which kernel and oracle may disagree about a fault, per search mode
WIDE_ENSEMBLE_BORDERLINE = {0: set(), 1: set()}


def test_wide_parameter_ensemble_vs_oracle_fp64():
    """BASELINE configs[4]'s parameter ranges (alpha in [0.0015, 0.015], n in [1.1, 3], Ksat in [0.01, 5]) under the
    synth_1 storm: kernel vs oracle per column; the same columns leave the reference's domain of validity."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 512
    E = W.ensemble_columns(N, seed=3)
    f = W.synth1_forcing()
    pr = np.repeat(f[:, 0:1], N, 1)
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(E["alpha"], E["n"], E["ksat"], E["theta_e"], E["theta_r"], E["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    for mode in (0, 1):
        eng = lg.LgarEngine(E["alpha"], E["n"], E["ksat"], E["theta_e"], E["theta_r"], E["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=torch.float64, search_mode=mode)
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff",), check=False)
        gst = eng.status.cpu().numpy()
        agree = ((st != 0) == (gst != 0))
        # Exact agreement is the rule.  The only disagreements admitted are the columns NAMED here: ones whose top layer
        # saturates (psi -> 0), where the reference's own decision (an isclose tie at 1e-8, the Se > 1 fault of insert_water)
        # hinges on the last bit of pow -- device vs glibc.  A new disagreement fails the test.
        flipped = set(int(i) for i in np.nonzero(~agree)[0])
        assert flipped <= WIDE_ENSEMBLE_BORDERLINE[mode], (mode, sorted(flipped))
        ok = (st == 0) & (gst == 0)
        assert ok.mean() > 0.5
        tot = eng.totals.cpu().numpy()
        assert _rel(tot[:8, ok], acc[:8, ok], 1e-3).max() <= 1e-6, mode
        got = out["runoff"].cpu().numpy()
        assert np.abs(got[:, ok] - ro[:, ok]).max() <= 1e-6 * max(1.0, np.abs(ro[:, ok]).max()), mode


@pytest.mark.timeout(900)
def test_wide_parameter_sweep_65536_columns_vs_oracle():
    """tools/parity_sweep.py's wide shape in the driver's suite: 65 536 columns drawn from configs[4]'s parameter ranges, every
    column against the oracle in the fast and the literal mode.  Saturating top layers make the reference's own decisions hinge on
    the last bit of pow there, so a few columns in 10^4 take another branch than the oracle under any
    other libm: the count is bounded (<= 0.05 % fault-flag flips, <= 0.05 % of the jointly valid columns off by more than
    1e-6; observed r03: 0.016-0.026 % in total), everything else agrees to 1e-6.  The mixed-precision mode is held to its own,
    wider bar on this ill-conditioned ensemble (<= 0.4 % flips)."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 65536
    E = W.ensemble_columns(N, seed=300)
    f = W.synth1_forcing()
    pr = np.repeat(f[:, 0:1], N, 1)
    pe = np.zeros_like(pr)
    ro, pc, acc, st = O.run_columns(E["alpha"], E["n"], E["ksat"], E["theta_e"], E["theta_r"], E["thickness"], pr, pe,
                                    pdm=0.0, dt_h=300.0 / 3600.0)
    scale = max(1.0, np.abs(ro).max())
    report = {}
    for mode in (1, 0, "mixed"):
        mk = dict(search_mode=1, geff_precision="f32") if mode == "mixed" else dict(search_mode=mode)
        eng = lg.LgarEngine(E["alpha"], E["n"], E["ksat"], E["theta_e"], E["theta_r"], E["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=torch.float64, **mk)
        out = eng.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff",), check=False)
        gst = eng.status.cpu().numpy() & 0x7f
        flips = int(((st != 0) != (gst != 0)).sum())
        both = (st == 0) & (gst == 0)
        col_err = np.abs(out["runoff"].cpu().numpy() - ro).max(0) / scale
        off = int((col_err[both] > 1e-6).sum())
        report[mode] = (flips, off)
        assert both.mean() > 0.9
        if mode == "mixed":
            assert flips <= 0.004 * N and off <= 0.02 * N, report
        else:
            assert flips <= 0.0005 * N and off <= 0.0005 * N, report
    print("wide sweep (flag flips, columns off by > 1e-6):", report)


def test_percolating_bottom_boundary_matches_oracle():
    """bottom_mode=1 (LGAR-C intent; the reference crashes there, so this is oracle-vs-kernel only): fronts that
    reach the domain bottom leave as percolation and the column keeps integrating."""
    import lgar_py_amd as lg
    from oracle import lgar_oracle as O
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    thick = [4.0, 4.0, 4.0]
    p = O.make_params(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], thick, pdm=0.0, dt_h=float(g["dt_h"]))
    p.bottom_mode = 1
    s = O.init_state(p)
    g0 = s.ending_volume  # the oracle's initial volume
    ref = O.run(p, s, g["forcing"][:, 0], g["forcing"][:, 1])
    assert ref["status"] == 0 and ref["acc"][:, 5].sum() > 0.05  # percolation happened
    for mode in (0, 1):
        eng = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], thick, n_columns=70,
                            dt_h=float(g["dt_h"]), ponded_depth_max=0.0, bottom_mode=1, search_mode=mode)
        v0 = float(eng.ending_volume[0])
        pr, pe = _forcing(g, 70)
        out = eng.forward(pr, pe, series=lg.ACC_NAMES)
        for j, nm in enumerate(lg.ACC_NAMES):
            got = out[nm][:, 0].cpu().numpy()
            assert np.abs(got - ref["acc"][:, j]).max() <= 1e-6 * max(1.0, np.abs(ref["acc"][:, j]).max()), (mode, nm)
        t = eng.totals[:, 0].cpu().numpy()
        assert abs(t[5] - ref["acc"][:, 5].sum()) <= 1e-9 and t[5] > 0.05
        # (this 12 cm toy column does not close its balance exactly: the reference clamps layer-0 fronts to the column
        #  depth, Layer.py:456-457, which drops their overshoot -- so the residual is held to the oracle's own, not to zero)
        acc = ref["acc"]
        res_ref = float(g0) + acc[:, 0].sum() - acc[:, 4].sum() - acc[:, 2].sum() - acc[-1, 8] - acc[:, 5].sum() - acc[-1, 9]
        res = v0 + t[0] - t[4] - t[2] - t[8] - t[5] - t[9]
        assert abs(res - res_ref) <= 1e-8 and abs(res) <= 0.1, (res, res_ref)


def test_capacity_chain_on_gpu_many_fronts():
    """A job large enough for the front-capacity chain to engage by itself (> 1024 waves): most columns stay within 8
    fronts, every 97th one grows to 31 (the reference's own trajectory, manyfronts_pulse_84) and moves through the 16- and
    32-slot kernels inside the same call; one more pulse series drives those past 32 -> LGAR_ST_OVERFLOW only there."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "manyfronts_pulse_84.npz"))
    N = 70_000
    T = g["forcing"].shape[0]
    scale = torch.zeros(N, dtype=torch.float64)
    scale[::97] = 1.0
    pr = torch.tensor(g["forcing"][:, 0:1]) * scale[None, :]
    pe = torch.zeros_like(pr)
    eng = _engine(g, N, torch.float64)
    out = eng.forward(pr, pe, series=("runoff", "infiltration", "ending_volume"), basin=("runoff",))
    big = torch.nonzero(scale).flatten()
    assert int(eng.n_fronts[big].min()) == 31 and int(eng.n_fronts.max()) == 31 and int(eng.n_fronts.min()) == 3
    for nm in ("runoff", "infiltration", "ending_volume"):
        ref = g["acc"][:, lg.ACC_NAMES.index(nm)]
        got = out[nm][:, big].cpu().numpy()
        assert _rel(got, ref[:, None]).max() <= 1e-6, nm
        assert (got == got[:, :1]).all()
    assert np.allclose(out["basin:runoff"].cpu().numpy(), g["acc"][:, 4] * len(big), rtol=1e-9, atol=1e-12)
    fr = eng.fronts()
    c = int(big[5])
    assert _rel(fr["depth"][:31, c], g["fronts"][T - 1, :31, 0]).max() <= 1e-6
    assert _rel(fr["theta"][:31, c], g["fronts"][T - 1, :31, 1]).max() <= 1e-6
    more = torch.tensor(np.tile([0.02, 0.0], 30)[:, None]) * scale[None, :]
    with pytest.raises(lg.LgarStatusError, match="front overflow"):
        eng.forward(more, torch.zeros_like(more))
    st = eng.status.cpu().numpy()
    assert (st[big.numpy()] == 8).all() and (np.delete(st, big.numpy()) == 0).all()
    # fp32 (the configuration bench.py measures): same structure, totals close to the reference
    e32 = _engine(g, N, torch.float32)
    e32.forward(pr, pe, series=())
    assert int(e32.n_fronts.max()) >= 28 and bool((e32.status == 0).all())
    assert abs(float(e32.totals[3, big[0]]) - g["acc"][:, 3].sum()) <= 5e-3 * g["acc"][:, 3].sum()


@pytest.mark.parametrize("name", ["phil_hourly_3000", "synth0_phil_1500", "four_layer_synth0_600", "bushland_hourly_1500"])
def test_single_step_transitions_from_injected_reference_states(name):
    """Per-branch state transitions: every column of ONE launch starts from a different state captured from the
    reference (written straight into the engine's state tensors) and takes one forward(); outputs must equal the
    reference's next step.  Covers create / insert_water / in-layer moves / base case / merge / layer crossing /
    dry-over-wet as (state_in, forcing) -> state_out pairs, and the HBM -> LDS state load path."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    nf = g["nfronts"]
    T = len(nf)
    ks = sorted(set([k for k in range(T - 1) if nf[k + 1] != nf[k]] + list(range(0, T - 1, 17))))
    N = len(ks)
    eng = _engine(g, N, torch.float64)
    F = min(eng.depth.shape[0], g["fronts"].shape[1])
    fr = np.transpose(g["fronts"][ks][:, :F, :], (2, 1, 0))  # [5, F, N]
    dev = eng.device
    for j, t in enumerate((eng.depth, eng.theta, eng.psi, eng.k, eng.dzdt)):
        t[:F].copy_(torch.tensor(fr[j], device=dev))
    lay = g["front_layer"][ks][:, :F].T.astype(np.int16)
    bot = g["front_bottom"][ks][:, :F].T.astype(np.int16)
    flags = np.where(lay >= 0, lay | (bot << 7), 0).astype(np.uint8)
    eng.flags[:F].copy_(torch.tensor(flags, device=dev))
    eng.n_fronts.copy_(torch.tensor(nf[ks].astype(np.int32), device=dev))
    eng.scalars[0].copy_(torch.tensor(g["acc"][ks, 8], device=dev))
    eng.scalars[1].copy_(torch.tensor(g["prev_precip"][ks], device=dev))
    eng.scalars[2].copy_(torch.tensor(g["acc"][ks, 9], device=dev))
    eng.scalars[3:8].copy_(torch.tensor(g["giuh_queue"][ks].T, device=dev))
    nxt = [k + 1 for k in ks]
    pr = torch.tensor(g["forcing"][nxt, 0][None, :])
    pe = torch.tensor(g["forcing"][nxt, 1][None, :])
    out = eng.forward(pr, pe, series=lg.ACC_NAMES)
    for j, nm in enumerate(lg.ACC_NAMES):
        got = out[nm][0].cpu().numpy()
        assert _rel(got, g["acc"][nxt, j]).max() <= 1e-6, nm
    res = eng.fronts()
    assert (res["n_fronts"] == nf[nxt]).all()
    for c, k in enumerate(nxt):
        n = int(nf[k])
        assert _rel(res["depth"][:n, c], g["fronts"][k, :n, 0]).max() <= 1e-6, k
        assert _rel(res["theta"][:n, c], g["fronts"][k, :n, 1]).max() <= 1e-6, k
        assert (res["layer"][:n, c] == g["front_layer"][k, :n]).all() and (res["to_bottom"][:n, c] == g["front_bottom"][k, :n]).all()


@pytest.mark.parametrize("lanes", [4, 6, 21, 64])
@pytest.mark.parametrize("name", ["phil_hourly_3000", "synth1_phil", "manyfronts_pulse_84", "five_layer_phil_500",
                                  "two_layer_synth1", "six_layer_synth1", "frozen07_phil_hourly_400"])
def test_cooperating_lanes_reproduce_one_lane_per_column(name, lanes):
    """LgarDims.forward_lanes: small fp64 jobs give every column 4..64 lanes (any number: 64 / lanes columns per wavefront,
    left-over lanes join the last group) that split the Geff trapezoid's nodes (and the pows that open it) between them, with
    one front table per group of lanes.  Every per-step output, the final front tables and the run totals are those of one
    lane per column BIT FOR BIT."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ncol = 5
    res = {}
    for k in (1, lanes):
        eng = _engine(g, ncol, torch.float64, forward_lanes=k)
        pr, pe = _forcing(g, ncol)
        out = eng.forward(pr, pe, series=lg.ACC_NAMES, check=False)
        res[k] = (out, eng)
    a, ea = res[1]
    b, eb = res[lanes]
    for nm in lg.ACC_NAMES:
        assert torch.equal(a[nm], b[nm]), nm
    for t in ("depth", "theta", "psi", "k", "dzdt", "flags", "n_fronts", "status", "scalars"):
        assert torch.equal(getattr(ea, t), getattr(eb, t)), t
    assert torch.equal(ea.totals, eb.totals)


@pytest.mark.parametrize("precision", ["native", "f32"])
@pytest.mark.parametrize("nint", [7, 24, 50, 121, 128])
@pytest.mark.parametrize("lanes", [4, 9, 12, 16, 33, 64])
def test_cooperating_lanes_other_interval_counts(nint, lanes, precision):
    """The cooperative trapezoid at interval counts other than the bundled 120: heads in batches of 16 (+ a tail), nodes and
    terms in rounds of two or four per lane (ragged last rounds), the sum in batches of 16 + 8 + a tail -- bit for bit one lane
    per column (three-layer and six-layer soil: riders and search evaluations of every layer count; groups of 12 lanes and more
    take two moving fronts at a time, odd group sizes leave the upper half one lane more, and with six layers a half of 6 or 8
    lanes is too small for the riders of a deep front: the one-front path takes over).  precision "f32": the same for the
    mixed-precision trapezoid, whose four-node groups (1 .. 31 of them, plus the leftover nodes) are split over the lanes."""
    import lgar_py_amd as lg
    for name in ("synth1_phil", "six_layer_synth1"):
        g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        g["nint"] = np.int64(nint)

        class G(dict):
            files = list(g)
        gg = G(g)
        res = {}
        for k in (1, lanes):
            eng = _engine(gg, 3, torch.float64, forward_lanes=k, geff_precision=precision)
            pr, pe = _forcing(gg, 3)
            out = eng.forward(pr, pe, series=lg.ACC_NAMES, check=False)
            res[k] = (out, eng)
        a, ea = res[1]
        b, eb = res[lanes]
        for nm in lg.ACC_NAMES:
            assert torch.equal(a[nm], b[nm]), (name, nm)
        for t in ("depth", "theta", "psi", "k", "dzdt", "flags", "n_fronts", "status", "totals"):
            assert torch.equal(getattr(ea, t), getattr(eb, t)), (name, t)


@pytest.mark.parametrize("lanes", [4, 6, 21, 64])
@pytest.mark.parametrize("name", ["phil_hourly_3000", "synth1_phil", "manyfronts_pulse_84", "five_layer_phil_500",
                                  "two_layer_synth1", "frozen07_phil_hourly_400"])
def test_cooperating_lanes_with_the_mixed_precision_trapezoid(name, lanes):
    """geff_precision="f32" on a small job: the lanes of a column split the four-node groups of the mixed-precision trapezoid
    (and the front sweep's evaluations) between them -- every per-step output, the final front tables and the run totals are
    those of the mixed-precision mode with one lane per column BIT FOR BIT."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ncol = 5
    res = {}
    for k in (1, lanes):
        eng = _engine(g, ncol, torch.float64, forward_lanes=k, geff_precision="f32")
        pr, pe = _forcing(g, ncol)
        out = eng.forward(pr, pe, series=lg.ACC_NAMES, check=False)
        res[k] = (out, eng)
    a, ea = res[1]
    b, eb = res[lanes]
    for nm in lg.ACC_NAMES:
        assert torch.equal(a[nm], b[nm]), nm
    for t in ("depth", "theta", "psi", "k", "dzdt", "flags", "n_fronts", "status", "scalars", "totals"):
        assert torch.equal(getattr(ea, t), getattr(eb, t)), t


def test_cooperating_lanes_on_distinct_columns_and_the_default_choice():
    """200 different columns: the library's own choice (64 lanes for this size) and a forced 7 reproduce one lane per column."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    N = 200
    P = W.perturbed_columns(N, seed=21)
    f = W.synth1_forcing()
    pr = torch.tensor(f[:, 0:1] * W.forcing_scale(N, seed=22)[None, :])
    pe = torch.zeros_like(pr)
    outs = []
    for k in (1, 0, 7):
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=torch.float64, forward_lanes=k)
        out = eng.forward(pr, pe, series=("runoff", "infiltration", "ending_volume"), check=False)
        outs.append((out, eng.status.clone(), eng.theta.clone(), eng.n_fronts.clone()))
    for o, st, th, nf in outs[1:]:
        for nm in ("runoff", "infiltration", "ending_volume"):
            assert torch.equal(o[nm], outs[0][0][nm]), nm
        assert torch.equal(st, outs[0][1]) and torch.equal(th, outs[0][2]) and torch.equal(nf, outs[0][3])
    assert int((outs[0][1] != 0).sum()) > 0  # the ensemble contains columns the reference faults on: same flags
