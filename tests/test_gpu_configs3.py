"""GPU (-m gpu): BASELINE configs[3] at its TOTAL size on the one GPU a box has -- 8 388 608 synthetic columns x 144 steps,
fp32, fast search -- through properties that do not depend on how the job is sharded:

* replicas: the job is 131 072 distinct columns x 64 scrambled copies; every copy gives bit-identical series, status and
  totals (a column's result does not depend on the wavefront, the block or the position it lands in);
* shard independence: the contiguous shard a rank of an 8-way split would own (workloads.shard_bounds), run on its own,
  reproduces exactly that slice of the whole job bit for bit, and the eight shards' basin-runoff vectors add up to the
  whole job's vector (the all-reduce of configs[3] is this sum);
* surface water balance on every column; the head of the job against the fp64 oracle at the fp32 tolerance of
  tests/test_gpu_parity.py.

(The RCCL exchange itself is exercised by tests/test_gpu_distributed.py with a group of one.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N_TOTAL = 8 * (1 << 20)
N_DISTINCT = 1 << 17
WORLD = 8


def _distinct():
    from lgar_py_amd import workloads as W
    return W.perturbed_columns(N_DISTINCT, seed=0), W.forcing_scale(N_DISTINCT, seed=1000), W.synth1_forcing()


def _engine(order_dev, lo, hi):
    """engine + forcing of columns [lo, hi) of the job (column j = distinct column order[j]); everything built on the device"""
    import lgar_py_amd as lg
    P, sc, f = _distinct()
    sel = order_dev[lo:hi]
    Q = {k: torch.tensor(v, device="cuda")[:, sel].contiguous() for k, v in P.items()}
    eng = lg.LgarEngine(Q["alpha"], Q["n"], Q["ksat"], Q["theta_e"], Q["theta_r"], Q["thickness"], dt_h=300.0 / 3600.0,
                        ponded_depth_max=0.0, dtype=torch.float32)
    pr = (torch.tensor(f[:, 0], device="cuda")[:, None] * torch.tensor(sc, device="cuda")[sel][None, :]).to(torch.float32).contiguous()
    return eng, pr, torch.zeros_like(pr)


@pytest.fixture(scope="module")
def whole_job():
    order = np.random.default_rng(17).permutation(N_TOTAL) % N_DISTINCT
    order_dev = torch.tensor(order, device="cuda")
    eng, pr, pe = _engine(order_dev, 0, N_TOTAL)
    out = eng.forward(pr, pe, series=("runoff",), basin=("runoff",), check=False)
    torch.cuda.synchronize()
    del pr, pe
    return dict(eng=eng, order=order, order_dev=order_dev, runoff=out["runoff"], basin=out["basin:runoff"])


@pytest.mark.timeout(1200)
def test_replicas_are_bitwise_equal_at_8m_columns(whole_job):
    order, eng = whole_job["order"], whole_job["eng"]
    first = np.full(N_DISTINCT, -1, dtype=np.int64)
    u, pos = np.unique(order, return_index=True)
    first[u] = pos
    assert (first >= 0).all()
    ref_pos = torch.tensor(first[order], device="cuda")
    ro = whole_job["runoff"]
    for t0 in range(0, ro.shape[0], 16):  # in slabs: a gathered copy of the whole series would double the 4.8 GB
        blk = ro[t0:t0 + 16]
        assert torch.equal(blk, blk[:, ref_pos]), t0
    assert torch.equal(eng.status, eng.status[ref_pos])
    assert torch.equal(eng.totals, eng.totals[:, ref_pos])
    assert torch.equal(eng.n_fronts, eng.n_fronts[ref_pos])
    assert float((eng.status == 0).float().mean()) > 0.8


@pytest.mark.timeout(1200)
def test_every_shard_of_an_eight_way_split_reproduces_its_slice(whole_job):
    from lgar_py_amd.workloads import shard_bounds
    eng = whole_job["eng"]
    total = torch.zeros_like(whole_job["basin"])
    for rank in range(WORLD):
        lo, hi = shard_bounds(N_TOTAL, WORLD, rank)
        e, pr, pe = _engine(whole_job["order_dev"], lo, hi)
        out = e.forward(pr, pe, series=("runoff",), basin=("runoff",), check=False)
        assert torch.equal(out["runoff"], whole_job["runoff"][:, lo:hi]), rank
        assert torch.equal(e.status, eng.status[lo:hi]) and torch.equal(e.totals, eng.totals[:, lo:hi]), rank
        total += out["basin:runoff"]
        del e, pr, pe, out
    # what the RCCL all-reduce of configs[3] computes: fp64 sums of per-wave fp32 partial sums, in any order
    scale = float(whole_job["basin"].abs().max())
    assert scale > 0 and float((total - whole_job["basin"]).abs().max()) <= 1e-9 * scale


def test_surface_water_balance_on_every_column(whole_job):
    eng = whole_job["eng"]
    ok = eng.status == 0
    tot = eng.totals.double()
    surface = tot[0] - tot[3] - tot[4] - tot[8]  # precipitation = infiltration + runoff + ponded water
    assert float(surface[ok].abs().max()) <= 1e-4
    ro = torch.zeros(N_TOTAL, dtype=torch.float64, device="cuda")
    for t0 in range(0, whole_job["runoff"].shape[0], 16):
        ro += whole_job["runoff"][t0:t0 + 16].double().sum(0)
    assert float(((ro - tot[4]).abs() / torch.clamp(tot[0], min=1.0))[ok].max()) <= 1e-5


def test_head_of_the_8m_job_matches_the_oracle(whole_job):
    from oracle import lgar_oracle as O
    order, eng = whole_job["order"], whole_job["eng"]
    P, sc, f = _distinct()
    n = 512
    sel = order[:n]
    pr = f[:, 0:1] * sc[sel][None, :]
    ro, pc, acc, st = O.run_columns(*(np.ascontiguousarray(P[k][:, sel]) for k in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")),
                                    pr, np.zeros_like(pr), pdm=0.0, dt_h=300.0 / 3600.0)
    gst = eng.status[:n].cpu().numpy()
    ok = (st == 0) & (gst == 0)
    assert ok.mean() > 0.5
    tot = eng.totals[:, :n].double().cpu().numpy()
    for j in (0, 3, 9):
        r = np.abs(tot[j] - acc[j])[ok] / np.maximum(np.abs(acc[j][ok]), 1e-2)
        assert np.percentile(r, 99) <= 5e-3 and r.max() <= 5e-2, (j, r.max())
    dro = np.abs(tot[4] - acc[4])[ok] / np.maximum(acc[0][ok], 1.0)
    assert np.percentile(dro, 99) <= 5e-3 and dro.max() <= 5e-2
