"""GPU (-m gpu): the dpLGAR(nn.Module) surface driven exactly like the reference's agent loop
(agents/DifferentiableLGAR.py:117-125): model(x[i]); mass_balance.change_mass(model)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_io import write_forcing, write_soil_dat

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cfg(tmp_path, g, data="Phillipsburg", models="shorter_subcycle", n=300, **over):
    from lgar_py_amd import config
    os.makedirs(tmp_path / "data", exist_ok=True)
    soil = write_soil_dat(str(tmp_path / "data" / "vG_default_params.dat"))
    step = 60 if models == "shorter_subcycle" else 5
    forcing = write_forcing(str(tmp_path / "data" / "forcing.csv"), g["forcing"][:n], step_min=step)
    ov = {"data.forcing_file": forcing, "data.soil_params_file": soil, "models.endtime": n * step / 60.0}
    ov.update(over)
    return config.load_config(data=data, models=models, cwd=str(tmp_path), overrides=ov)


def test_drop_in_agent_loop_single_column(tmp_path):
    from lgar_py_amd.data import Data
    from lgar_py_amd.model import MassBalance, dpLGAR
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    n = 300
    cfg = _cfg(tmp_path, g, n=n)
    data = Data(cfg)
    assert len(data) == n
    model = dpLGAR(cfg)
    assert len(model.alpha) == 3 and model.alpha[0].dim() == 0 and model.alpha[0].requires_grad
    assert isinstance(model.alpha, torch.nn.ParameterList) and len(list(model.parameters())) == 9
    assert cfg.data.soil_index["theta_e"] == 1
    mb = MassBalance(cfg, model)
    assert abs(float(mb.starting_volume) - float(g["init_volume"])) < 1e-9
    for i in range(n):
        x, y = data[i]
        runoff, perc = model(x)
        assert runoff.dim() == 0
        for j, nm in enumerate(["precip", "PET", "AET", "infiltration", "runoff", "percolation", "giuh_runoff", "discharge"]):
            ref = g["acc"][i, j]
            assert abs(float(getattr(model, nm)) - ref) <= 1e-6 * max(abs(ref), 1e-6), (i, nm)
        assert abs(float(model.ending_volume) - g["acc"][i, 9]) <= 1e-6 * g["acc"][i, 9]
        mb.change_mass(model)
        assert float(model.AET) == 0.0
    err = mb.report_mass(model, log=lambda s: None)
    assert abs(float(mb.AET) - g["acc"][:n, 2].sum()) <= 1e-6 * g["acc"][:n, 2].sum()
    assert abs(float(mb.AET) - 2.782323715146482) <= 1e-6 * 2.78  # SURVEY §8c anchor: 300 Phillipsburg steps
    assert abs(float(err)) < 1e-6
    fronts = model.wetting_fronts()
    assert len(fronts) == int(g["nfronts"][n - 1]) == model.calc_num_wetting_fronts()
    assert abs(fronts[0]["theta"] - g["fronts"][n - 1, 0, 1]) <= 1e-6
    # epoch reset (agents/DifferentiableLGAR.py:105-107)
    model.set_internal_states()
    mb.reset_mass(model)
    assert model.calc_num_wetting_fronts() == 3 and float(model.ending_volume) == float(mb.starting_volume)
    r0, _ = model(data[0][0])
    assert abs(float(model.precip) - g["acc"][0, 0]) <= 1e-12


def test_batched_forward_and_ensemble(tmp_path):
    from lgar_py_amd import workloads as W
    from lgar_py_amd.model import dpLGAR
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    cfg = _cfg(tmp_path, g, data="synth_1", models="five_minute", n=144)
    N = 130
    model = dpLGAR(cfg, n_columns=N)
    x = torch.tensor(g["forcing"])[:, None, :].expand(144, N, 2)
    runoff, perc = model(x)  # T steps in one launch
    assert runoff.shape == (144, N)
    ref = torch.tensor(g["acc"][:, 4], device=runoff.device)[:, None]
    assert float((runoff - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    assert abs(float(model.runoff[7]) - g["acc"][:, 4].sum()) <= 1e-9
    # per-column ensemble: [N, L] overrides become one [N] Parameter per layer
    P = W.perturbed_columns(N, seed=5)
    ens = dpLGAR(cfg, n_columns=N, alpha=P["alpha"].T, n=P["n"].T, ksat=P["ksat"].T, theta_e=P["theta_e"].T,
                 theta_r=P["theta_r"].T)
    assert ens.alpha[0].shape == (N,)
    try:
        ens(x)
    except ValueError:
        pass  # a perturbed column may leave the reference's domain of validity (it raises there too)
    assert float(ens.infiltration.std()) > 0
    # update_soil_parameters pushes new parameter values to the device without resetting the state
    with torch.no_grad():
        model.ksat[0].mul_(0.5)
    model.update_soil_parameters()
    assert abs(float(model.engine.ksat[0, 0]) - 0.225) < 1e-12


def test_streamed_run_matches_reference_golden():
    """the streamed driver (pinned host -> side-stream expansion -> chunked launches, basin sums from the
    kernel epilogue) on the bundled 3000-hour Phillipsburg forcing against the accumulators the reference produced, step
    by step; and against the oracle on heterogeneous columns."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from lgar_py_amd.pipeline import run_streamed
    from oracle import lgar_oracle as O
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    x = g["forcing"]
    T, N = x.shape[0], 192
    kw = dict(dt_h=1.0, ponded_depth_max=2.0, dtype=torch.float64)
    eng = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], n_columns=N, **kw)
    names = ("runoff", "AET", "infiltration", "ending_volume", "giuh_runoff")
    basin = run_streamed(eng, x, chunk=411, series=names)
    for nm in names:
        ref = g["acc"][:, lg.ACC_NAMES.index(nm)]
        got = basin[nm].cpu().numpy() / N
        assert np.abs(got - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-6), nm
    for j in range(8):
        assert abs(float(eng.totals[j, 0]) - g["acc"][:, j].sum()) <= 1e-6 * max(abs(g["acc"][:, j].sum()), 1e-6)
    # heterogeneous columns + per-column forcing scale + area weights: basin series == oracle's weighted column sum
    N2, T2 = 128, 600
    P = W.perturbed_columns(N2, seed=51)
    sc = W.forcing_scale(N2, 0.5, 1.5, seed=52)
    w = np.random.default_rng(53).random(N2)
    pr = x[:T2, 0:1] * sc[None, :]
    pe = x[:T2, 1:2] * np.ones((1, N2))
    ro, pc, acc, st = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr, pe, pdm=2.0, dt_h=1.0)
    e2 = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], **kw)
    ok = st == 0  # columns the reference would raise on carry weight 0
    got = run_streamed(e2, x[:T2], scale=sc, chunk=128, series=("runoff",), weights=w * ok, check=False)["runoff"].cpu().numpy()
    assert ((e2.status.cpu().numpy() != 0) == (st != 0)).all()
    want = (ro * (w * ok)[None, :]).sum(1)
    assert np.abs(got - want).max() <= 1e-6 * max(want.max(), 1e-6)


def test_streamed_run_equals_single_shot():
    """pipeline.run_streamed (double-buffered chunks on a side stream) == one launch over the whole [T, N] forcing."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from lgar_py_amd.pipeline import run_streamed
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    N, T = 300, 700
    P = W.perturbed_columns(N, seed=31)
    sc = W.forcing_scale(N, 0.5, 1.5, seed=32)
    kw = dict(dt_h=1.0, ponded_depth_max=2.0, dtype=torch.float64)
    x = g["forcing"][:T]
    a = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], **kw)
    pr = torch.tensor(x[:, 0:1] * sc[None, :])
    pe = torch.tensor(x[:, 1:2] * np.ones((1, N)))
    full = a.forward(pr, pe, series=("runoff", "AET"), check=False)
    b = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], **kw)
    got = run_streamed(b, x, scale=sc, chunk=97, series=("runoff", "AET"), reduce_basin=False)
    for nm in ("runoff", "AET"):
        assert torch.equal(got[nm], full[nm]), nm
    assert torch.allclose(a.totals, b.totals, rtol=1e-13, atol=1e-15) and torch.equal(a.theta, b.theta)
    c = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], **kw)
    basin = run_streamed(c, x, scale=sc, chunk=256, series=("runoff",))["runoff"]
    assert basin.shape == (T,)
    assert torch.allclose(basin, full["runoff"].sum(1), rtol=1e-12, atol=1e-12)


def test_streamed_per_column_forcing_from_a_mapped_file_equals_single_shot(tmp_path):
    """pipeline.run_streamed_columns: [T, N] forcing with N distinct columns in a memory-mapped file -> pinned double buffer
    -> HBM on a side stream, chunk by chunk == one launch over the whole resident [T, N] forcing, bit for bit; per-column PET
    from a second file, a basin PET series, and an fp32 file feeding an fp64 engine (converted on the device)."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from lgar_py_amd.pipeline import open_forcing_file, run_streamed_columns, write_forcing_file
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    N, T = 517, 700
    rng = np.random.default_rng(5)
    P = W.perturbed_columns(N, seed=31)
    kw = dict(dt_h=1.0, ponded_depth_max=2.0, dtype=torch.float64)
    x = g["forcing"][:T]
    pr = x[:, 0:1] * rng.uniform(0.5, 2.5, (T, N))   # every column its own series
    pe = x[:, 1:2] * rng.uniform(0.8, 1.2, (T, N))
    mk = lambda: lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], **kw)
    a = mk()
    full = a.forward(torch.tensor(pr), torch.tensor(pe), series=("runoff", "AET"), check=False)
    fp = open_forcing_file(write_forcing_file(str(tmp_path / "precip.npy"), pr))
    fe = open_forcing_file(write_forcing_file(str(tmp_path / "pet.npy"), pe))
    assert isinstance(fp, np.memmap) and fp.shape == (T, N)
    b = mk()
    st = {}
    got = run_streamed_columns(b, fp, fe, chunk=97, series=("runoff", "AET"), reduce_basin=False, check=False, stats=st)
    for nm in ("runoff", "AET"):
        assert torch.equal(got[nm], full[nm]), nm
    assert torch.equal(a.theta, b.theta) and torch.equal(a.status, b.status)
    assert st["bytes_host_to_device"] == 2 * T * N * 8 and st["chunks"] == 8 and st["host_to_device_GBps"] > 0
    # basin sums, basin PET series
    c = mk()
    full2 = c.forward(torch.tensor(pr), torch.tensor(np.repeat(x[:, 1:2], N, 1)), series=("runoff",), check=False)
    d = mk()
    basin = run_streamed_columns(d, fp, x[:, 1], chunk=256, series=("runoff",), check=False)["runoff"]
    assert basin.shape == (T,) and torch.allclose(basin, full2["runoff"].sum(1), rtol=1e-12, atol=1e-12)
    assert torch.equal(c.theta, d.theta)
    # an fp32 file into the fp64 engine: the same as uploading the rounded values
    f32 = open_forcing_file(write_forcing_file(str(tmp_path / "precip32.npy"), pr.astype(np.float32)))
    e, f = mk(), mk()
    want = e.forward(torch.tensor(pr.astype(np.float32).astype(np.float64)), torch.zeros(T, N, dtype=torch.float64),
                     series=("runoff",), check=False)
    got32 = run_streamed_columns(f, f32, None, chunk=300, series=("runoff",), reduce_basin=False, check=False)
    assert torch.equal(got32["runoff"], want["runoff"])
    with pytest.raises(ValueError):
        run_streamed_columns(mk(), pr[:, :5], None)
    # the three ways a chunk leaves the host give the same bits: pread out of the file (above), numpy copies out of a plain
    # array, and a map REGISTERED with the runtime (no staging copy); and the per-step series streamed BACK into host files --
    # one registered (the copy engines write the page cache), one plain (pinned double buffer + writer thread)
    from lgar_py_amd.pipeline import close_forcing_file, create_forcing_file
    assert st["source"].startswith("pread")
    st2 = {}
    got2 = run_streamed_columns(mk(), np.array(fp), np.array(fe), chunk=97, series=("runoff",), reduce_basin=False, check=False,
                                stats=st2, reader_threads=3)
    assert st2["source"].startswith("numpy") and torch.equal(got2["runoff"], full["runoff"])
    rp, re_ = open_forcing_file(str(tmp_path / "precip.npy"), register=True), open_forcing_file(str(tmp_path / "pet.npy"), register=True)
    o1 = open_forcing_file(create_forcing_file(str(tmp_path / "runoff_out.npy"), (T, N), "float64"), register=True)
    o2 = open_forcing_file(create_forcing_file(str(tmp_path / "aet_out.npy"), (T, N), "float64"), writable=True)
    try:
        st3 = {}
        h = mk()
        basin3 = run_streamed_columns(h, rp, re_, chunk=97, series=("runoff",), check=False, stats=st3,
                                      host_out={"runoff": o1, "AET": o2})["runoff"]
        assert st3["source"].startswith("registered") and st3["bytes_device_to_host"] == 2 * T * N * 8
        assert np.array_equal(np.asarray(o1), full["runoff"].cpu().numpy()) and np.array_equal(np.asarray(o2), full["AET"].cpu().numpy())
        assert torch.allclose(basin3, full["runoff"].sum(1), rtol=1e-12, atol=1e-12) and torch.equal(h.theta, a.theta)
    finally:
        for m_ in (rp, re_, o1):
            close_forcing_file(m_)
        del rp, re_, o1, o2

    class FailingSource:  # a forcing source whose read fails part way (a truncated file, an I/O error): surfaces, never hangs
        ndim, shape, dtype = 2, pr.shape, pr.dtype

        def __getitem__(self, sl):
            if sl.start >= 300:
                raise OSError("read failed at row %d" % sl.start)
            return pr[sl]

    with pytest.raises(OSError, match="read failed"):
        run_streamed_columns(mk(), FailingSource(), None, chunk=97, series=("runoff",), check=False)


def test_in_kernel_basin_aggregation():
    """LgarStepOut.basin: per-step basin sums reduced in the kernel epilogue (wave reduction + fp64 atomics) equal the
    column sums of the per-step series, with and without weights, including a ragged tail wave."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    from oracle import lgar_oracle as O
    for N in (70, 1000):
        P = W.perturbed_columns(N, seed=41)
        sc = W.forcing_scale(N, 0.5, 1.0, seed=42)
        f = W.synth1_forcing()
        pr = torch.tensor(f[:, 0:1] * sc[None, :])
        pe = torch.zeros_like(pr)
        w = torch.rand(N, dtype=torch.float64)
        # the checker: the oracle's per-column runoff, summed over columns (weighted / unweighted), valid columns only
        ro, _, _, ost = O.run_columns(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], pr.numpy(),
                                      pe.numpy(), pdm=0.0, dt_h=300.0 / 3600.0)
        okw = torch.tensor((ost == 0).astype(np.float64))
        e0 = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                           ponded_depth_max=0.0, dtype=torch.float64)
        b0 = e0.forward(pr, pe, series=(), basin=("runoff",), weights=w * okw, check=False)["basin:runoff"].cpu().numpy()
        want = (ro * (w * okw).numpy()[None, :]).sum(1)
        assert np.abs(b0 - want).max() <= 1e-6 * want.max(), N
        for dtype, tol in ((torch.float64, 1e-13), (torch.float32, 1e-6)):
            eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"],
                                dt_h=300.0 / 3600.0, ponded_depth_max=0.0, dtype=dtype)
            out = eng.forward(pr, pe, series=("runoff", "infiltration"), basin=("runoff", "infiltration", "discharge"),
                              check=False)
            for nm in ("runoff", "infiltration"):
                ref = out[nm].double().sum(1)
                assert torch.allclose(out["basin:" + nm], ref, rtol=tol, atol=tol * float(ref.abs().max())), (N, dtype, nm)
            assert out["basin:discharge"].shape == (144,) and float(out["basin:discharge"].sum()) > 0
            eng.reset()
            outw = eng.forward(pr, pe, series=("runoff",), basin=("runoff",), weights=w, check=False)
            refw = (outw["runoff"].double() * w.to(outw["runoff"].device)[None, :]).sum(1)
            assert torch.allclose(outw["basin:runoff"], refw, rtol=tol * 10, atol=tol * 10 * float(refw.abs().max()))


def test_basin_sums_two_ways():
    """LgarStepOut.basin: where the series is stored, the basin sum is a second, deterministic pass over it
    (lgar_basin_reduce_kernel: same bits on every run); without a series the forward kernels reduce it themselves (atomics).
    Both equal the weighted column sum of the series; LgarEngine.forward puts a scratch series behind a basin-only request."""
    import lgar_py_amd as lg
    from lgar_py_amd import workloads as W
    for N, dtype, tol in ((5000, torch.float64, 1e-13), (9001, torch.float32, 1e-6), (777, torch.float64, 1e-13)):  # 16-byte rows or not
        P = W.perturbed_columns(N, seed=5)
        sc = W.forcing_scale(N, 0.5, 1.0, seed=6)
        f = W.synth1_forcing()
        pr = torch.tensor(f[:, 0:1] * sc[None, :])
        pe = torch.zeros_like(pr)
        w = torch.rand(N, dtype=torch.float64)
        eng = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                            ponded_depth_max=0.0, dtype=dtype)
        a = eng.forward(pr, pe, series=("runoff",), basin=("runoff", "infiltration"), weights=w, check=False)
        ref = (a["runoff"].double() * w.to(a["runoff"].device).to(dtype).double()[None, :]).sum(1)
        scale = float(ref.abs().max())
        assert scale > 0 and float((a["basin:runoff"] - ref).abs().max()) <= 10 * tol * scale
        eng.reset()
        b = eng.forward(pr, pe, series=("runoff",), basin=("runoff", "infiltration"), weights=w, check=False)
        assert torch.equal(a["basin:runoff"], b["basin:runoff"]) and torch.equal(a["basin:infiltration"], b["basin:infiltration"])
        assert eng._basin_scratch[0] == 144 and list(eng._basin_scratch[1]) == ["infiltration"]
        # the same request with no buffer behind it: the forward kernels' own reduction
        eng2 = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                             ponded_depth_max=0.0, dtype=dtype, basin_scratch_bytes=0)
        c = eng2.forward(pr, pe, series=(), basin=("runoff", "infiltration"), weights=w, check=False)
        assert eng2._basin_scratch[1] == {}
        for nm in ("runoff", "infiltration"):
            s = float(a["basin:" + nm].abs().max())
            assert float((c["basin:" + nm] - a["basin:" + nm]).abs().max()) <= 10 * tol * s, (N, nm)
        # the cap is over ALL names of an engine: room for one series only -> the second name falls back to the kernels' own sums
        one = 144 * N * (8 if dtype == torch.float64 else 4)
        eng3 = lg.LgarEngine(P["alpha"], P["n"], P["ksat"], P["theta_e"], P["theta_r"], P["thickness"], dt_h=300.0 / 3600.0,
                             ponded_depth_max=0.0, dtype=dtype, basin_scratch_bytes=one)
        d = eng3.forward(pr, pe, series=(), basin=("runoff", "infiltration"), weights=w, check=False)
        assert list(eng3._basin_scratch[1]) == ["runoff"]
        for nm in ("runoff", "infiltration"):
            s = float(a["basin:" + nm].abs().max())
            assert float((d["basin:" + nm] - a["basin:" + nm]).abs().max()) <= 10 * tol * s, (N, nm)
        eng3.release_scratch()
        assert eng3._basin_scratch[1] == {}


def test_forcing_broadcast_on_gpu():
    """LgarDims.forcing_columns: [T, 1] basin forcing for every column (and [T, N/2]) equals replicated forcing bitwise."""
    import lgar_py_amd as lg
    g = np.load(os.path.join(GOLDEN, "phil_hourly_3000.npz"))
    f = torch.tensor(g["forcing"][:400])
    N = 200
    kw = dict(n_columns=N, dt_h=1.0, ponded_depth_max=2.0, dtype=torch.float64)
    a = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], **kw)
    full = a.forward(f[:, 0:1].expand(-1, N).contiguous(), f[:, 1:2].expand(-1, N).contiguous(), series=("runoff", "AET"))
    b = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], **kw)
    one = b.forward(f[:, 0:1], f[:, 1:2], series=("runoff", "AET"))
    for nm in full:
        assert torch.equal(full[nm], one[nm]), nm
    assert torch.equal(a.theta, b.theta)
    ref = torch.tensor(g["acc"][:400, 2], device=one["AET"].device)
    assert float((one["AET"][:, 7] - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    with pytest.raises(lg.LgarError, match="dividing"):
        b.forward(f[:, 0:1].expand(-1, 3).contiguous(), f[:, 1:2].expand(-1, 3).contiguous())
    # forcing_group: G consecutive columns share a forcing column
    sc = torch.linspace(0.5, 1.5, 50, dtype=torch.float64)
    pr, pe = f[:, 0:1] * sc[None, :], f[:, 1:2].expand(-1, 50).contiguous()
    c = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], **kw)
    grp = c.forward(pr, pe, series=("runoff",), forcing_group=4)["runoff"]
    d = lg.LgarEngine(g["alpha"], g["n"], g["ksat"], g["theta_e"], g["theta_r"], g["thickness"], **kw)
    rep = d.forward(pr.repeat_interleave(4, dim=1), pe.repeat_interleave(4, dim=1), series=("runoff",))["runoff"]
    assert torch.equal(grp, rep)
    with pytest.raises(lg.LgarError, match="dividing"):
        c.forward(pr, pe, forcing_group=3)
