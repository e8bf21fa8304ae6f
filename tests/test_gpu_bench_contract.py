"""GPU (-m gpu): bench.py's output contract (one JSON line with the driver's keys, roofline and cpu_baseline objects),
exercised on a small workload."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    env = dict(os.environ, LGAR_CPU_THREADS="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--columns", "8192"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "column-timesteps/sec" and d["unit"] == "column-timesteps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # bound names what limits the kernel (vector-ALU issue: VERDICT r04 asked for the honest label); achieved / peak / frac are
    # the HBM figures of the contract, and the compute side's busy fraction sits beside them at top level
    assert r["bound"] == "valu-issue" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "valu_busy_frac" in d
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 4 and c["value"] > 0 and "sample" in c
    units = 8192 * 144 * 2
    assert abs(d["value"] - units / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    assert d["faulted_columns"] == 0
