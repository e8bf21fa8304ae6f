"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer legs over the two CPU builds of the path's arithmetic -- the oracle
(oracle/lgar_oracle.c, gcc) and the DEVICE CODE compiled for the host (tests/devsim, clang) -- on the reference's crash
fixtures (the fault paths: NaN heads, negative pow bases, a front at the domain bottom), a trajectory per search / precision
mode and the tangent lanes.  SURVEY.md section 5 ("race detection / sanitizers": the reference has none; the build runs its CPU
restatements under them).  Each library is built once (cached beside its plain twin) and loaded by a child process running
under LD_PRELOAD of the matching sanitizer runtime; any finding aborts the child (-fno-sanitize-recover=all)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run_child(which, runtime, extra_env):
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONFAULTHANDLER="0", OMP_NUM_THREADS="2", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(HERE, "_sanitizer_child.py"), which], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "SANITIZER_CHILD_OK" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-6000:])


def test_oracle_under_asan_ubsan():
    runtime = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(runtime):
        pytest.skip("gcc has no AddressSanitizer runtime here")
    _run_child("oracle", runtime, {"LGAR_ORACLE_SANITIZE": "1"})


def test_device_code_on_the_host_under_asan_ubsan():
    sys.path.insert(0, HERE)
    import devsim
    runtime = devsim.sanitizer_runtime()
    if runtime is None:
        pytest.skip("the ROCm clang has no AddressSanitizer runtime here")
    _run_child("devsim", runtime, {"DEVSIM_SANITIZE": "1"})
