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


def _launch(world, N, out, **extra_env):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
        if world == 1 and not extra_env:
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
def test_two_ranks_shared_parameter_training_equals_single_process(tmp_path):
    """SURVEY section 8e, "training with shared parameters adds an all-reduce of L x 3 gradient scalars"
    (agents/DifferentiableLGAR.py:94-172): two ranks share the basin's 96 columns (and the one GPU), exchange the [T] runoff sums
    and the 9 parameter gradients every epoch, and after two epochs hold bit-equal parameters that equal the single-process
    run's to 1e-12."""
    N, epochs = 96, 2

    def launch(world, out):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.pop("WORLD_SIZE", None)
            env.pop("RANK", None)
            if world > 1:
                env.update(RANK=str(r), WORLD_SIZE=str(world))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_agent_worker.py"), str(tmp_path), out,
                                           str(N), str(epochs)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        for p in procs:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0, o[-3000:]

    launch(1, str(tmp_path / "one_%d.npz"))
    launch(2, str(tmp_path / "two_%d.npz"))
    one = np.load(tmp_path / "one_0.npz")
    parts = [np.load(tmp_path / ("two_%d.npz" % r)) for r in range(2)]
    assert not bool(one["sharded"]) and all(bool(p["sharded"]) and bool(p["in_sync"]) for p in parts)
    assert (int(parts[0]["lo"]), int(parts[0]["hi"]), int(parts[1]["lo"]), int(parts[1]["hi"])) == (0, 48, 48, 96)
    assert np.array_equal(parts[0]["params"].view(np.int64), parts[1]["params"].view(np.int64))
    for p in parts:
        assert np.abs(p["params"] - one["params"]).max() <= 1e-12, np.abs(p["params"] - one["params"]).max()
        assert np.allclose(p["loss"], one["loss"], rtol=1e-12, atol=0)
    assert np.isfinite(one["loss"]).all() and len(one["loss"]) == epochs


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


@pytest.mark.timeout(900)
def test_bench_under_the_drivers_launcher_with_four_ranks():
    """The driver's own command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with N = 4 ranks sharing the one GPU (gloo rehearsal, 4 096 columns
    per rank).  Four, not eight: the pool allows six processes on a card and this test process is one of them; the world-size-8
    sharding and all-reduce run on the CPU (tests/test_distributed_gloo.py).  One JSON line from rank 0, n_gpus 4, the
    collective named, four ranks' worth of work, every rank on a device that exists."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LGAR_DIST_BACKEND="gloo", LGAR_CPU_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LGAR_FORCE_DIST", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2",
                        "--warmup", "1", "--columns", "4096"], capture_output=True, text=True, env=env, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["columns_per_gpu"] == 4096
    assert "gloo all-reduce" in d["config"]["collective"] and "group of 4" in d["config"]["collective"]
    assert d["config"]["parallelism"] == "columns sharded x4" and d["faulted_columns"] == 0
    units = 4 * 4096 * 144 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    assert "sub_records" not in d and "cpu_baseline" not in d  # N > 1: rank 0 prints the headline only


@pytest.mark.timeout(900)
def test_rccl_group_of_one_runs_the_real_all_reduce(tmp_path):
    """configs[3]'s exchange on what one GPU can host: the worker joins an RCCL ("nccl") process group of world size 1 and,
    with LGAR_FORCE_DIST=1, distributed.basin_runoff runs dist.all_reduce on the DEVICE [T] vector (no host hop).  The result
    equals the run without a process group bit for bit (a sum over one rank)."""
    N = 1000
    _launch(1, N, str(tmp_path / "plain_%d.npz"))
    _launch(1, N, str(tmp_path / "rccl_%d.npz"), LGAR_TEST_BACKEND="nccl", LGAR_FORCE_DIST="1")
    a, b = np.load(tmp_path / "plain_0.npz"), np.load(tmp_path / "rccl_0.npz")
    assert np.array_equal(a["runoff"], b["runoff"], equal_nan=True) and np.array_equal(a["status"], b["status"])
    assert np.array_equal(a["basin"], b["basin"]) and b["basin"].sum() > 0


@pytest.mark.timeout(900)
def test_bench_takes_the_rccl_path_with_one_gpu():
    """`LGAR_FORCE_DIST=1 python bench.py --gpus 1`: a fresh child initialises backend "nccl" (RCCL) with world size 1
    before any GPU call, and every pass all-reduces the device [T] vector and crosses the barriers of the timed region --
    bench.py's N > 1 code path, executed."""
    env = dict(os.environ, LGAR_FORCE_DIST="1", LGAR_CPU_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LGAR_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--columns", "65536", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "nccl all-reduce" in d["config"]["collective"] and "group of 1" in d["config"]["collective"]
    assert d["faulted_columns"] == 0 and d["basin_runoff_total_cm"] > 0
    # the same job without the group: same basin total (the all-reduce over one rank is the identity)
    env.pop("LGAR_FORCE_DIST")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--columns", "65536", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=800)
    assert q.returncode == 0, q.stderr[-3000:]
    e = json.loads([ln for ln in q.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert e["config"]["collective"] is None
    assert abs(e["basin_runoff_total_cm"] - d["basin_runoff_total_cm"]) <= 1e-9 * abs(e["basin_runoff_total_cm"])


def test_a_failing_rank_stops_its_siblings_instead_of_hanging():
    """bench.py --gpus 2 whose rank 1 cannot start (backend nccl, one visible GPU): the spawner reports the failure and stops
    rank 0 instead of leaving it in the rendezvous until the RCCL timeout."""
    import time
    env = dict(os.environ, LGAR_DIST_BACKEND="nccl")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LGAR_FORCE_DIST"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--columns", "4096", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert p.returncode != 0 and "only 1 visible" in (p.stderr + p.stdout)
    assert time.time() - t0 < 200
