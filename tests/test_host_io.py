"""CPU: host-side config composition and input readers (no compute)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def write_soil_dat(path, rows=18):
    from lgar_py_amd import data as D
    te = {12: 0.4513, 13: 0.4773, 14: 0.4617}
    tr = {12: 0.0648, 13: 0.0831, 14: 0.0668}
    with open(path, "w") as f:
        f.write("Texture\t        theta_r\ttheta_e\talpha(cm^-1)\tn\tm\tKs(cm/h)\n")
        for i in range(rows):
            a, n, k = D.VG_TABLE[i]
            f.write('"T-%d"  \t\t%g \t%g\t%g \t%g\t%g\t%g\n' % (i, tr.get(i, 0.05), te.get(i, 0.4), a, n, 1 - 1 / n, k))
    return path


def write_forcing(path, x_cm_per_h, hash_header=False, step_min=60):
    with open(path, "w") as f:
        f.write(("#" if hash_header else "") + "Time,P(mm/h),PET(mm/h)\n")
        for i, (p, e) in enumerate(x_cm_per_h):
            f.write("2016-10-01 %02d:%02d:00,%r,%r\n" % ((i * step_min) // 60 % 24, (i * step_min) % 60, float(p) * 10.0, float(e) * 10.0))
        if hash_header:
            f.write("\n\n")
    return path


def test_config_keys_and_time_derivations(tmp_path):
    from lgar_py_amd import config
    cfg = config.load_config(cwd=str(tmp_path))
    assert cfg.models.nsteps == 3000 and cfg.models.num_subcycles == 1 and cfg.models.subcycle_length_h == 1.0
    assert cfg.data.layer_soil_type == [12, 13, 14] and cfg.data.ponded_depth_max == 2
    assert cfg.data.forcing_file.startswith(str(tmp_path)) and cfg.constants.nint == 120
    cfg2 = config.load_config(data="synth_1", models="five_minute", overrides={"models.endtime": 6.0})
    assert cfg2.models.nsteps == 72 and abs(cfg2.models.subcycle_length_h - 300 / 3600) < 1e-15
    cfg3 = config.load_config(models="base")
    assert cfg3.models.num_subcycles == 12
    cfg.data.soil_index = {"m": 4}
    assert cfg.data.soil_index.m == 4


def test_soil_table_and_forcing_readers(tmp_path):
    from lgar_py_amd import data as D
    soil = D.read_soil_table(write_soil_dat(str(tmp_path / "soil.dat")))
    assert len(soil["Texture"]) == 18 and soil["Texture"][12] == "T-12"
    assert soil["theta_e"][13] == 0.4773 and soil["theta_r"][14] == 0.0668
    a, n, k = D.read_test_params()
    assert float(a[12]) == 0.0031297 and float(n[13]) == 1.299 and float(k[14]) == 0.45
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    p = write_forcing(str(tmp_path / "f.txt"), g["forcing"], hash_header=True, step_min=5)
    times, x = D.read_forcing(p)
    assert len(times) == 144 and np.allclose(x, g["forcing"], rtol=0, atol=1e-15)
    _, x2 = D.read_forcing(p, nsteps=10)
    assert x2.shape == (10, 2)
    assert abs(D.calculate_nse(np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 3.0])) - 1.0) < 1e-15


def test_forcing_csvs_to_mapped_files(tmp_path):
    """pipeline.forcing_csvs_to_files: one CSV per column in the reference's two on-disk formats -> the [T, N] files that
    run_streamed_columns maps (mm/h -> cm/h, data/Data.py:37); short files are refused."""
    from lgar_py_amd.pipeline import forcing_csvs_to_files, open_forcing_file
    rng = np.random.default_rng(0)
    T, N = 30, 5
    P, E = rng.uniform(0, 20, (T, N)), rng.uniform(0, 1, (T, N))
    paths = []
    for c in range(N):
        p = tmp_path / ("col%d.csv" % c)
        with open(p, "w") as f:
            f.write(("#Time" if c % 2 else "Time") + ",P(mm/h),PET(mm/h)\n")
            for t in range(T):
                f.write("2020-01-01 %02d:00:00,%r,%r\n" % (t % 24, float(P[t, c]), float(E[t, c])))
            if c % 2:
                f.write("\n\n")  # the synth files end in blank lines
        paths.append(str(p))
    got = forcing_csvs_to_files(paths, str(tmp_path / "p.npy"), str(tmp_path / "e.npy"), nsteps=24, dtype="float64",
                                block_columns=2)  # (blocks of 2, 2 and 1 columns)
    assert got == (24, N)
    fp, fe = open_forcing_file(str(tmp_path / "p.npy")), open_forcing_file(str(tmp_path / "e.npy"))
    assert isinstance(fp, np.memmap) and fp.shape == (24, N)
    assert np.allclose(fp, P[:24] * 0.1, rtol=0, atol=1e-15) and np.allclose(fe, E[:24] * 0.1, rtol=0, atol=1e-15)
    with open(paths[2], "w") as f:
        f.write("Time,P(mm/h),PET(mm/h)\n2020-01-01 00:00:00,1.0,0.1\n")
    with pytest.raises(ValueError, match="forcing rows"):
        forcing_csvs_to_files(paths, str(tmp_path / "p2.npy"), str(tmp_path / "e2.npy"))
    # a failed conversion leaves nothing behind: neither partial outputs nor their temporaries
    assert not [f for f in os.listdir(tmp_path) if f.startswith(("p2.npy", "e2.npy"))]


def test_chunk_bounds_fixed_size_and_schedule():
    """pipeline.chunk_bounds: what run_streamed_columns cuts a [T, N] series into -- a fixed row count, or a schedule whose last
    entry repeats; every row exactly once, in order, no empty chunk."""
    from lgar_py_amd.pipeline import chunk_bounds
    assert chunk_bounds(144, 16) == [(lo, lo + 16) for lo in range(0, 144, 16)]
    assert chunk_bounds(100, 97) == [(0, 97), (97, 100)]
    assert chunk_bounds(144, (4, 12, 32, 48)) == [(0, 4), (4, 16), (16, 48), (48, 96), (96, 144)]
    assert chunk_bounds(10, (3,)) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert chunk_bounds(5, (8, 2)) == [(0, 5)]
    assert chunk_bounds(0, 5) == []
    rng = np.random.default_rng(1)
    for _ in range(200):
        T = int(rng.integers(0, 400))
        sched = [int(v) for v in rng.integers(1, 60, int(rng.integers(1, 6)))]
        b = chunk_bounds(T, sched)
        assert [lo for lo, _ in b] == [0] * (T > 0) + [hi for _, hi in b[:-1]]
        assert all(hi > lo for lo, hi in b) and (not b or b[-1][1] == T)
        assert all(hi - lo == sched[min(i, len(sched) - 1)] for i, (lo, hi) in enumerate(b[:-1]))
    for bad in (0, -3, (), (4, 0)):
        with pytest.raises(ValueError):
            chunk_bounds(10, bad)
