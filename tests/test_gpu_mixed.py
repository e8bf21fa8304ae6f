"""GPU (-m gpu): the mixed-precision mode (geff_precision="f32", LgarDims.geff_mode = 1) through the C-ABI: fp64 column
state, branches and mass bookkeeping, fp64 heads and end nodes of the Geff trapezoid, its 119 interior nodes with the fp32
HARDWARE transcendentals (v_log_f32 / v_exp_f32), summed in fp64.

Against the reference's golden vectors, against the oracle and against the native fp64 kernels.  The bars (derived in
tests/test_devsim_golden.py and DESIGN.md section 4): front tables 1e-6; per-step outputs within 2e-5 of the water moving
through the column in that step; run totals 2e-6 of max(|total|, total rainfall); 1e-3 relative on every single per-step
value as a backstop; IDENTICAL fault flags."""
import os

import numpy as np
import pytest

from conftest import check_fault_kind, GOLDEN
from test_devsim_golden import MIXED_FLUX, MIXED_TOTAL, mixed_mode_check
from test_gpu_parity import TRAJ, _engine, _forcing, _rel

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("mode", [1, 2], ids=["fast_search", "fast_capacity_chain"])
@pytest.mark.parametrize("name", TRAJ)
def test_mixed_precision_trajectory_vs_reference_golden(name, mode):
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ncol = 67
    crash = int(g["crash_step"])
    T = crash if crash >= 0 else g["forcing"].shape[0]
    eng = _engine(g, ncol, torch.float64, search_mode=mode, geff_precision="f32")
    pr, pe = _forcing(g, ncol, slice(0, T))
    out = eng.forward(pr, pe, series=lg.ACC_NAMES)
    got = np.stack([out[nm].cpu().numpy() for nm in lg.ACC_NAMES], 2)  # [T, ncol, NACC]
    assert (got == got[:, :1]).all(), "replicated columns must be bit-identical"
    mixed_mode_check(got[:, 0], g["acc"][:T], T)
    fr = eng.fronts()
    nf = int(g["nfronts"][T - 1])
    assert (fr["n_fronts"] == nf).all()
    assert _rel(fr["depth"][:nf, 0], g["fronts"][T - 1, :nf, 0]).max() <= 1e-6
    assert _rel(fr["theta"][:nf, 0], g["fronts"][T - 1, :nf, 1]).max() <= 1e-6
    assert (fr["layer"][:nf, 0] == g["front_layer"][T - 1, :nf]).all()
    assert (fr["to_bottom"][:nf, 0] == g["front_bottom"][T - 1, :nf]).all()
    if crash >= 0:
        pr1, pe1 = _forcing(g, ncol, slice(T, T + 1))
        with pytest.raises(lg.LgarStatusError):
            eng.forward(pr1, pe1)
        assert bool((eng.status != 0).all())
        check_fault_kind(g, eng.status.cpu().numpy())


def test_mixed_geff_leaf_on_the_hardware():
    """The mixed trapezoid with the real v_log_f32 / v_exp_f32 against the reference's literal trapezoid (leaf op 6) over the
    ranges the call sites see: <= 1e-6 relative for every input (measured: max 6e-7, median 2e-8, mean +5e-8 -- the hardware's
    v_log_f32 / v_exp_f32 do not round symmetrically; with the host's libm: max 1.6e-7, median 8e-9, mean 0)."""
    import lgar_py_amd as lg
    rng = np.random.default_rng(0)
    n = 1 << 16
    alpha, nn = rng.uniform(0.003, 0.009, n), rng.uniform(1.25, 1.75, n)
    kw = dict(alpha=alpha, n=nn, ksat=np.full(n, 0.3), theta_e=np.full(n, 0.46), theta_r=np.full(n, 0.07))
    theta = lambda h: 0.07 + 0.39 * (1 + (alpha * h) ** nn) ** -(1 - 1 / nn)
    h1 = 10 ** rng.uniform(0.5, 3.3, n)
    for t1, t2 in ((theta(h1), np.full(n, 0.46)), (theta(h1), theta(h1 * 10 ** rng.uniform(-2, -0.05, n))),
                   (theta(h1), theta(h1 * rng.uniform(0.7, 0.98, n)))):
        ref = lg.leaf_batch("geff_literal", t1, t2, **kw).cpu().numpy()
        mix = lg.leaf_batch("geff_mixed", t1, t2, **kw).cpu().numpy()
        e = (mix - ref) / ref
        assert np.abs(e).max() <= 1e-6 and np.median(np.abs(e)) <= 5e-8 and abs(e.mean()) <= 1e-7, (np.abs(e).max(), e.mean())


def test_mixed_precision_ensemble_vs_oracle_and_native_fp64():
    """131 072 columns of the bench ensemble: the mixed mode flags exactly the columns the native fp64 kernels flag, and on
    every other column the run totals agree to 2e-6 of the column's water input (measured: median 1e-9, max 7e-7); the first
    2 048 columns against the oracle: same flags, same bars."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    N = 1 << 17
    P = W.perturbed_columns(N, seed=0)
    sc = W.forcing_scale(N, seed=1000)
    f = W.synth1_forcing()
    res = {}
    for label, kw in (("native", {}), ("mixed", {"geff_precision": "f32"})):
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=torch.float64, **kw)
        pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[None, :]).contiguous()
        out = eng.forward(pr, torch.zeros_like(pr), series=("runoff",), check=False)
        res[label] = (eng.totals.clone(), eng.status.clone(), eng.n_fronts.clone(), out["runoff"])
    tn, sn, nn_, rn = res["native"]
    tm, sm, nm_, rm = res["mixed"]
    assert torch.equal(sn, sm) and torch.equal(nn_, nm_)
    ok = sn == 0
    scale = torch.maximum(tn[:8].abs(), tn[0:1]).clamp_min(1e-2)
    assert float(((tm[:8] - tn[:8]).abs() / scale)[:, ok].max()) <= MIXED_TOTAL
    # per-step runoff: at a front event one step's flux moves by up to ~100x the Geff noise, so over
    # 19 M column-steps the worst one is not the typical one (measured: worst 1.6e-5 cm on a 0.22 cm/step scale)
    d = (rm - rn).abs()[:, ok]
    flux = float(rn.abs().max())
    assert float(d.max()) <= 2e-4 * flux and float((d > 2e-6 * flux).double().mean()) <= 1e-5
    n = 2048
    pr = f[:, 0:1] * sc[None, :n]
    ro, pc, acc, st = O.run_columns(*(np.ascontiguousarray(P[k][:, :n]) for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")),
                                    pr, np.zeros_like(pr), pdm=0.0, dt_h=300.0 / 3600.0)
    assert np.array_equal(st != 0, sm[:n].cpu().numpy() != 0)
    good = st == 0
    tot = tm[:8, :n].cpu().numpy()
    sc8 = np.maximum(np.maximum(np.abs(acc[:8]), acc[0:1]), 1e-2)
    assert (np.abs(tot - acc[:8]) / sc8)[:, good].max() <= MIXED_TOTAL
    assert np.abs(rm[:, :n].cpu().numpy() - ro)[:, good].max() <= 2e-4 * np.abs(ro).max()


def test_mixed_precision_replicas_are_bitwise_equal():
    """A column's result does not depend on its wavefront in the mixed mode either: 4096 distinct columns x 64 scrambled
    copies give bit-identical series."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    n = 1 << 18
    order = np.random.default_rng(3).permutation(n) % 4096
    P = W.perturbed_columns(4096, seed=5)
    sc = W.forcing_scale(4096, seed=6)
    f = W.synth1_forcing()
    Q = {k: np.ascontiguousarray(v[:, order]) for k, v in P.items()}
    eng = lg.LgarEngine(Q["alpha"], Q["n"], Q["ksat"], Q["theta_e"], Q["theta_r"], Q["thickness"], dt_h=300.0 / 3600.0,
                        ponded_depth_max=0.0, dtype=torch.float64, geff_precision="f32")
    pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc[order], device="cuda")[None, :]).contiguous()
    one = eng.forward(pr, torch.zeros_like(pr), series=("runoff",), check=False)["runoff"]
    first = np.full(4096, -1, dtype=np.int64)
    u, pos = np.unique(order, return_index=True)
    first[u] = pos
    ref_pos = torch.tensor(first[order], device="cuda")
    assert torch.equal(one, one[:, ref_pos]) and torch.equal(eng.status, eng.status[ref_pos])
    assert torch.equal(eng.totals, eng.totals[:, ref_pos])


def test_mixed_precision_is_an_fp64_fast_mode_option():
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    P = W.PHILLIPSBURG
    args = [P[k] for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")]
    with pytest.raises(lg.LgarError):
        lg.LgarEngine(*args, n_columns=4, dtype=torch.float32, geff_precision="f32")
    with pytest.raises(lg.LgarError):
        lg.LgarEngine(*args, n_columns=4, dtype=torch.float64, search_mode=0, geff_precision="f32")
    with pytest.raises(lg.LgarError):
        lg.LgarEngine(*args, n_columns=4, geff_precision="half")


@pytest.mark.parametrize("name", ["rand03", "synth2_phil", "bench_col10707", "five_layer_synth1", "manyfronts_pulse_84"])
def test_mixed_giuh_queue_in_place_equals_queue_in_registers(name):
    """The one-lane mixed-precision kernels (MODE 3) keep the GIUH queue in the column's rows of the `scalars` array and update it
    there (lgar_device.hpp, Column::GIUH_MEM); the cooperating-lanes kernel (MODE 6) carries it in registers like every other
    kernel.  Same additions in the same order: routed runoff, discharge, the stored queue and every other series must agree bit
    for bit -- in one launch, and when the run is cut into launches that each start from a queue left in memory by the last
    (the flag "something is queued" is then rebuilt from the loaded rows).  The routed runoff itself is checked against the
    reference's golden series."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    T = g["forcing"].shape[0]
    ncol = 130  # two full waves and a ragged one
    pr, pe = _forcing(g, ncol)
    runs = {}
    for tag, lanes, cuts in (("in_place", 1, (T,)), ("registers", 64, (T,)), ("in_place_cut", 1, (7, 31, 32, T))):
        eng = _engine(g, ncol, torch.float64, geff_precision="f32", forward_lanes=lanes)
        parts, t0 = [], 0
        for t1 in cuts:
            if t1 <= t0:
                continue
            out = eng.forward(pr[t0:t1].contiguous(), pe[t0:t1].contiguous(), series=lg.ACC_NAMES)
            parts.append(np.stack([out[nm].cpu().numpy() for nm in lg.ACC_NAMES], 2))
            t0 = t1
        runs[tag] = (np.concatenate(parts, 0), eng.scalars.cpu().numpy().copy(), eng.status.cpu().numpy().copy())
    ref = runs["in_place"]
    assert (ref[2] == 0).all()
    for tag in ("registers", "in_place_cut"):
        got = runs[tag]
        assert np.array_equal(got[2], ref[2]), tag
        assert np.array_equal(got[0], ref[0]), "%s: series differ from the one-launch in-place run" % tag
        assert np.array_equal(got[1], ref[1]), "%s: ponded water / previous precipitation / ending volume / queue rows differ" % tag
    q = lg.ACC_NAMES.index("giuh_runoff")
    assert (ref[0][:, :, q] == ref[0][:, :1, q]).all()
    assert (ref[0][:, 0, q] > 0).sum() > 3  # the fixture routes runoff
    mixed_mode_check(ref[0][:, 0], g["acc"][:T], T)
