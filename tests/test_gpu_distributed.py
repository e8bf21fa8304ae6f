"""GPU (-m gpu): the multi-rank path with the REAL HIP engine.  A one-GPU box cannot host an RCCL group of two devices, so
the two ranks form a gloo group and share the GPU (LGAR_DIST_BACKEND=gloo is bench.py's rehearsal mode for the same
reason); the code path -- ShardedColumns + LgarEngine + one all-reduce of the basin runoff [T] -- is the one that runs
with backend "nccl" (= RCCL over xGMI) on an 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, N, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if world == 1:
            env.pop("WORLD_SIZE")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(N), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o[-3000:]


@pytest.mark.timeout(900)
def test_two_ranks_sharing_one_gpu_equal_single_rank(tmp_path):
    """the same inputs sharded 1 and 2 ways give identical per-column outputs (bitwise) and the
    same reduced basin runoff up to summation order."""
    N = 1000  # ragged shards and ragged tail waves
    _launch(1, N, str(tmp_path / "one_%d.npz"))
    _launch(2, N, str(tmp_path / "two_%d.npz"))
    one = np.load(tmp_path / "one_0.npz")
    parts = [np.load(tmp_path / ("two_%d.npz" % r)) for r in range(2)]
    assert int(parts[0]["lo"]) == 0 and int(parts[0]["hi"]) == int(parts[1]["lo"]) == 500 and int(parts[1]["hi"]) == N
    ro = np.concatenate([p["runoff"] for p in parts], axis=1)
    assert np.array_equal(ro, one["runoff"], equal_nan=True)
    assert np.array_equal(np.concatenate([p["status"] for p in parts]), one["status"])
    for p in parts:  # every rank holds the all-reduced [T] vector
        assert np.allclose(p["basin"], one["basin"], rtol=1e-12, atol=1e-12)
    assert one["basin"].sum() > 0


@pytest.mark.timeout(900)
def test_bench_self_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (gloo rehearsal on the one GPU) and reports
    n_gpus = 2 with twice the single-rank work; `--gpus 2` can therefore never silently measure one GPU."""
    env = dict(os.environ, LGAR_DIST_BACKEND="gloo", LGAR_CPU_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--columns", "16384"], capture_output=True, text=True, env=env, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["columns_per_gpu"] == 16384
    assert "gloo" in d["config"]["collective"]
    units = 2 * 16384 * 144 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    # a launcher that disagrees with --gpus is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--columns", "4096"], capture_output=True, text=True, env=env2, timeout=300)
    assert q.returncode != 0 and "WORLD_SIZE" in (q.stderr + q.stdout)
