"""Inputs of the hot path: soil table, hard-coded vG parameter table, forcing series.

Formats follow the reference's bundled files: forcing CSV `Time,P(mm/h),PET(mm/h)` (the
synth files use `#Time` and end in blank lines; /root/reference/data/forcing_data_synth_1.txt), whitespace
`.dat` soil tables with quoted texture names (/root/reference/data/vG_default_params.dat).  The loaders emit
[T, N] column-fastest tensors ready for the kernels.
"""
import csv
import shlex

import numpy as np
import torch

# (alpha [1/cm], n, Ksat [cm/h]) by soil type: the values dpLGAR(cfg) takes its parameters from
# (/root/reference/dpLGAR/data/utils.py:108-180, read_test_params); theta_e / theta_r come from the .dat file.
VG_TABLE = np.array([
    (0.01, 1.25, 0.612), (0.02, 1.42, 0.3348), (0.01, 1.47, 0.504), (0.03, 1.75, 4.32), (0.04, 3.18, 26.64),
    (0.03, 1.21, 0.468), (0.02, 1.33, 0.54), (0.03, 1.45, 1.584), (0.01, 1.68, 1.836), (0.02, 1.32, 0.432),
    (0.01, 1.52, 0.468), (0.01, 1.66, 0.756), (0.0031297, 1.6858, 0.45), (0.0083272, 1.299, 0.07),
    (0.0037454, 1.6151, 0.45), (0.009567, 1.3579, 0.07), (0.005288, 1.5276, 0.02), (0.004467, 1.4585, 0.2),
])


def read_test_params(cfg=None):
    """(alpha, n, Ksat) fp64 tensors of the 18-row table (data/utils.py:108)."""
    t = torch.tensor(VG_TABLE, dtype=torch.float64)
    return t[:, 0].clone(), t[:, 1].clone(), t[:, 2].clone()


def read_soil_table(path):
    """Whitespace-delimited soil table -> dict of numpy columns (Texture, theta_r, theta_e, alpha, n, m, Ks).
    Row i is soil type i (0-based), as the reference indexes it (data/utils.py:66-67)."""
    rows = []
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    for ln in lines[1:]:
        parts = shlex.split(ln)
        rows.append((parts[0],) + tuple(float(v) for v in parts[1:7]))
    cols = list(zip(*rows))
    return dict(Texture=list(cols[0]), theta_r=np.array(cols[1]), theta_e=np.array(cols[2]), alpha=np.array(cols[3]),
                n=np.array(cols[4]), m=np.array(cols[5]), Ks=np.array(cols[6]))


def read_forcing(path, nsteps=None, mm_to_cm=0.1):
    """Forcing file -> (times, x[T, 2]) with x = (precip, PET) in cm/h (data/Data.py:32-37)."""
    times, p, e = [], [], []
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = [h.strip().lstrip("#") for h in next(rd)]
        it, ip, ie = header.index("Time"), header.index("P(mm/h)"), header.index("PET(mm/h)")
        for row in rd:
            if not row or not "".join(row).strip():
                continue
            times.append(row[it])
            p.append(float(row[ip]))
            e.append(float(row[ie]))
            if nsteps is not None and len(times) >= nsteps:
                break
    x = np.stack([np.array(p), np.array(e)], axis=1) * mm_to_cm
    return times, x


def tile_forcing(x, n_columns, scale=None, device="cuda:0", dtype=torch.float64):
    """x[T, 2] -> (precip[T, N], pet[T, N]) on the device, optionally scaled per column."""
    x = torch.as_tensor(x, dtype=torch.float64, device=device)
    T = x.shape[0]
    s = torch.ones(n_columns, dtype=torch.float64, device=device) if scale is None else \
        torch.as_tensor(scale, dtype=torch.float64, device=device)
    precip = (x[:, 0:1] * s[None, :]).to(dtype).contiguous()
    pet = (x[:, 1:2] * s[None, :]).to(dtype).contiguous() if scale is not None else \
        x[:, 1:2].expand(T, n_columns).to(dtype).contiguous()
    return precip, pet


def read_observations(path, nsteps=None, column=None):
    """Observation series of cfg.data.observations (data/Data.py:59-65 reads the `total_precipitation` column of that
    CSV): `column` if given, else `total_precipitation`, else the first numeric column that is not the time stamp.
    Values are taken as they are (the reference applies no unit conversion)."""
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = [h.strip().lstrip("#") for h in next(rd)]
        rows = [r for r in rd if r and "".join(r).strip()]
    if column is None:
        column = "total_precipitation" if "total_precipitation" in header else None
    if column is not None:
        if column not in header:
            raise ValueError("observations file %s has no column %r (columns: %s)" % (path, column, header))
        j = header.index(column)
    else:
        j = None
        for c, nm in enumerate(header):
            if nm.lower() in ("time", "date", "datetime"):
                continue
            try:
                float(rows[0][c])
                j = c
                break
            except (ValueError, IndexError):
                continue
        if j is None:
            raise ValueError("observations file %s has no numeric column" % path)
    vals = np.array([float(r[j]) for r in rows[: nsteps if nsteps is not None else len(rows)]])
    return vals


class Data(torch.utils.data.Dataset):
    """Counterpart of dpLGAR.data.Data (data/Data.py:22-57): (x[2], y) per forcing row.  y holds the observations of
    cfg.data.observations when that key names a file (optional cfg.data.observation_column), else zeros -- the reference
    has its read_observations call commented out and trains against torch.rand (Data.py:42-43)."""

    def __init__(self, cfg):
        super().__init__()
        self.times, x = read_forcing(cfg.data.forcing_file, cfg.models.nsteps, cfg.conversions.mm_to_cm)
        self.x = torch.tensor(x, dtype=torch.float64)
        self.timestep_map = dict(enumerate(self.times))
        obs = cfg.data.get("observations") if hasattr(cfg.data, "get") else getattr(cfg.data, "observations", None)
        if obs:
            y = read_observations(obs, self.x.shape[0], cfg.data.get("observation_column") if hasattr(cfg.data, "get") else None)
            if len(y) < self.x.shape[0]:
                raise ValueError("observations file %s has %d rows, the forcing has %d" % (obs, len(y), self.x.shape[0]))
            self.y = torch.tensor(y, dtype=torch.float64)
        else:
            self.y = torch.zeros(self.x.shape[0], dtype=torch.float64)

    def __getitem__(self, i):
        return self.x[i], self.y[i]

    def __len__(self):
        return self.x.shape[0]


def calculate_nse(modeled, observed):
    """Nash-Sutcliffe efficiency (data/metrics.py:4-8)."""
    modeled, observed = np.asarray(modeled), np.asarray(observed)
    var = np.sum((observed - observed.mean()) ** 2)
    if var == 0.0:  # constant observations (e.g. none given): NSE is undefined
        return float("nan")
    return 1 - np.sum((observed - modeled) ** 2) / var
