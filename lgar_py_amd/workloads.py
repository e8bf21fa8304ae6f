"""Synthetic workloads of BASELINE.json's configs, generated without touching the
reference tree (the GPU box has no /root/reference).

The synth_1 forcing shape is restated from /root/reference/data/forcing_data_synth_1.txt
(144 rows @ 5 min; P = 20 mm/h for rows 6-10 and 59-128, else 0; PET = 0) and checked against the
golden fixture in tests/test_capi_host.py (test_workloads_match_fixtures).  Soil parameters are the Phillipsburg P-1..3 rows
(/root/reference/dpLGAR/data/utils.py:123-125,146-148,170-172; data/vG_default_params.dat:14-16).
"""
import numpy as np

PHILLIPSBURG = dict(
    alpha=[0.0031297, 0.0083272, 0.0037454],
    n=[1.6858, 1.299, 1.6151],
    ksat=[0.45, 0.07, 0.45],
    theta_e=[0.4513, 0.4773, 0.4617],
    theta_r=[0.0648, 0.0831, 0.0668],
    thickness=[44.0, 131.0, 25.0],
)
GIUH = (0.06, 0.51, 0.28, 0.12, 0.03)
PARAM_KEYS = ("alpha", "n", "ksat", "theta_e", "theta_r")


def synth1_forcing(tile=1):
    """[144*tile, 2] (precip, PET) in cm/h: data/forcing_data_synth_1.txt x mm_to_cm (data/Data.py:37)."""
    p = np.zeros(144)
    p[6:11] = 20.0
    p[59:129] = 20.0
    f = np.stack([p * 0.1, np.zeros(144)], axis=1)
    return np.tile(f, (tile, 1))


def perturbed_columns(n_columns, frac=0.10, seed=0, base=PHILLIPSBURG):
    """Per-column soils: base parameters x U(1-frac, 1+frac), fp64 arrays [L, N] (config 3 of SURVEY §8d)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in PARAM_KEYS:
        b = np.asarray(base[k], dtype=np.float64)[:, None]
        out[k] = b * (1.0 + frac * (2.0 * rng.random((len(base[k]), n_columns)) - 1.0))
    out["thickness"] = np.repeat(np.asarray(base["thickness"], dtype=np.float64)[:, None], n_columns, axis=1)
    return out


def forcing_scale(n_columns, lo=0.5, hi=1.5, seed=1):
    """Per-column forcing multiplier U(lo, hi) (config 3 option), so columns branch differently."""
    rng = np.random.default_rng(seed)
    return lo + (hi - lo) * rng.random(n_columns)


def ensemble_columns(n_columns, seed=0):
    """Config 5: per-column (alpha, n, Ksat)[3] ~ U(lb, ub) within models/config/shorter_subcycle.yaml:23-32
    (n in [1.1, 3], Ksat in [0.01, 5]); theta_e/theta_r/thickness from Phillipsburg."""
    rng = np.random.default_rng(seed)
    L = 3
    out = dict(
        alpha=0.0015 + (0.015 - 0.0015) * rng.random((L, n_columns)),
        n=1.1 + (3.0 - 1.1) * rng.random((L, n_columns)),
        ksat=0.01 + (5.0 - 0.01) * rng.random((L, n_columns)),
    )
    for k in ("theta_e", "theta_r", "thickness"):
        out[k] = np.repeat(np.asarray(PHILLIPSBURG[k], dtype=np.float64)[:, None], n_columns, axis=1)
    return out


def shard_bounds(n_total, world_size, rank):
    """Contiguous column shard [lo, hi) of rank: sizes differ by at most one."""
    q, r = divmod(n_total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)
