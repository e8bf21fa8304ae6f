"""CPU: bench.py's bookkeeping (algorithmic bytes of SURVEY §8d, committed-PMC lookups, CPU-baseline leg on a tiny sample)."""
import importlib.util
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    b = _bench()
    # SURVEY §8d's formula B_alg = 4 s + (L 6 s + 2 F_MAX (5 s + 1) + 2 16 s) / T with L = 3, F_MAX = 16; the survey
    # rounds B_col to 840 B (fp32) / 1648 B (fp64), the formula itself gives 872 / 1712
    assert abs(b.alg_bytes_per_col_step(4, 144) - (16 + 872 / 144)) < 1e-12
    assert abs(b.alg_bytes_per_col_step(4, 144) - 21.8) < 0.3 and abs(b.alg_bytes_per_col_step(4, 3000) - 16.3) < 0.05
    assert abs(b.alg_bytes_per_col_step(8, 144) - (32 + 1712 / 144)) < 1e-12 and abs(b.alg_bytes_per_col_step(8, 144) - 43.4) < 0.6


def test_committed_pmc_lookup(monkeypatch):
    """roofline.traffic comes from a committed profile only when that profile was taken on exactly this workload AND with this
    library build (content fingerprint); otherwise it is null with the reason."""
    b = _bench()
    alg = b.alg_bytes_per_col_step(4, 144) * 1048576 * 144
    t, why = b.measured_traffic(12345, 144, "f32")
    assert t is None and "no committed profile" in why
    rec = b._matching_profile(1048576, 144, "f32")
    assert rec is not None and rec["source"].startswith("profiles/r0")
    # as if the profile had been taken with this very build: the figure is reported and stays within 15 % of the 3.33 GB
    # algorithmic figure (no wasted re-reads)
    monkeypatch.setattr(b, "library_fingerprint", lambda: rec.get("library_fingerprint"))
    t, why = b.measured_traffic(1048576, 144, "f32")
    assert t is not None and 1.5e9 < t < 1.15 * alg and rec["source"] in why
    v = b.measured_valu(1048576, 144, "f32")
    assert v is not None and 0.5 < v["busy_frac"] <= 1.0 and v["source"] == rec["source"]
    # ... and with any other build it is not
    monkeypatch.setattr(b, "library_fingerprint", lambda: "0" * 64)
    t, why = b.measured_traffic(1048576, 144, "f32")
    assert t is None and "another build" in why and b.measured_valu(1048576, 144, "f32") is None


def test_cpu_baseline_leg_runs_on_a_tiny_sample(monkeypatch):
    b = _bench()
    monkeypatch.setenv("LGAR_CPU_THREADS", "2")
    r = b.cpu_baseline(target_s=0.3)
    assert r["kind"] == "port" and r["cores"] == 2 and r["value"] > 1e3 and "columns x 144 steps" in r["sample"]


def test_wait_ranks_stops_the_siblings_of_a_failed_rank():
    """a rank that dies early must not leave the others blocked until the RCCL timeout."""
    import subprocess
    import sys
    import time
    b = _bench()
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; print('rank0 up', flush=True); time.sleep(120)"],
                              stdout=subprocess.PIPE, text=True),
             subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(3)"])]
    out, failed = b.wait_ranks(procs)
    assert failed == (1, 3) and time.time() - t0 < 30
    assert all(p.poll() is not None for p in procs) and "rank0 up" in out
    procs = [subprocess.Popen([sys.executable, "-c", "print('{\"ok\": 1}')"], stdout=subprocess.PIPE, text=True),
             subprocess.Popen([sys.executable, "-c", "pass"])]
    out, failed = b.wait_ranks(procs)
    assert failed is None and out.strip() == '{"ok": 1}'
