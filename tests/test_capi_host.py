"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lgar.h declares;
struct layouts of the ctypes binding match the header; the product path refuses to run without a GPU
(no CPU fallback) and never imports the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "lgar.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lgar_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from lgar_py_amd import _capi
    lib = _capi.load()
    names = _declared_functions()
    assert "lgar_forward" in names and "lgar_state_init" in names and len(names) >= 7
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(_capi.EXPORTS) == names
    assert lib.lgar_fmax() == _capi.FMAX and lib.lgar_lmax() == _capi.LMAX
    assert b"gfx950" in lib.lgar_version()


def test_struct_layout_matches_header(tmp_path):
    """Compile a tiny C program against include/lgar.h and compare sizeof/offsetof with ctypes."""
    from lgar_py_amd import _capi
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "lgar.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %d %d %d %d\\n", sizeof(LgarDims), offsetof(LgarDims, dt_h),'
        ' offsetof(LgarDims, giuh), offsetof(LgarDims, iter_cap), sizeof(LgarParams), sizeof(LgarState),'
        ' sizeof(LgarForcing), sizeof(LgarStepOut), LGAR_FMAX, LGAR_LMAX, LGAR_GMAX, LGAR_NSCAL);return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    D = _capi.LgarDims
    want = [C.sizeof(D), D.dt_h.offset, D.giuh.offset, D.iter_cap.offset, C.sizeof(_capi.LgarParams),
            C.sizeof(_capi.LgarState), C.sizeof(_capi.LgarForcing), C.sizeof(_capi.LgarStepOut), _capi.FMAX,
            _capi.LMAX, _capi.GMAX, _capi.NSCAL]
    assert got == want


def test_no_cpu_fallback():
    import torch
    import lgar_py_amd as lg
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lg.LgarError, match="no CPU fallback"):
        lg.LgarEngine([1e-2] * 3, [1.5] * 3, [1.0] * 3, [0.4] * 3, [0.1] * 3, [10.0] * 3, n_columns=2)
    with pytest.raises(lg.LgarError, match="no CPU fallback"):
        lg.leaf_batch("geff", [0.1], [0.2], alpha=[0.01], n=[1.5], ksat=[1.0], theta_e=[0.4], theta_r=[0.1])


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under lgar_py_amd/ may reference it."""
    pk = os.path.join(ROOT, "lgar_py_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)
    code = "import sys; sys.path.insert(0, %r); import lgar_py_amd; assert not any('oracle' in m for m in sys.modules), 'oracle imported'" % ROOT
    subprocess.check_call([sys.executable, "-c", code])


def test_workloads_match_fixtures():
    from lgar_py_amd import workloads as W
    g = np.load(os.path.join(GOLDEN, "synth1_phil.npz"))
    assert np.array_equal(W.synth1_forcing(), g["forcing"])
    assert W.synth1_forcing(3).shape == (432, 2)
    for k in W.PARAM_KEYS:
        assert np.array_equal(np.asarray(W.PHILLIPSBURG[k]), g[k])
    P = W.perturbed_columns(1000, seed=0)
    for k in W.PARAM_KEYS:
        r = P[k] / np.asarray(W.PHILLIPSBURG[k])[:, None]
        assert r.min() >= 0.9 and r.max() <= 1.1 and r.std() > 0.03
    assert (P["theta_e"] > P["theta_r"]).all() and (P["n"] > 1.0).all()
    s = W.forcing_scale(1000)
    assert s.min() >= 0.5 and s.max() <= 1.5
    for n, w in ((10, 3), (8_000_000, 8), (7, 8)):
        b = [W.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_integration_md_stub_matches_the_ctypes_mirror():
    """INTEGRATION.md shows the binding a maintainer of the reference would add: its struct field lists must be the ones
    lgar_py_amd/_capi.py (checked against include/lgar.h above) uses."""
    import re
    from lgar_py_amd import _capi
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    i, j = txt.index("class LgarDims"), txt.index("class LgarParams")
    assert re.findall(r'\("(\w+)"', txt[i:j]) == [f[0] for f in _capi.LgarDims._fields_]
    for cls in ("LgarState", "LgarForcing", "LgarStepOut"):
        line = txt[txt.index("class " + cls):]
        line = line[:line.index("\n")]
        m = re.search(r"for k in \(([^)]*)\)", line)
        doc = [x.strip().strip('"') for x in m.group(1).split(",")] if m else re.findall(r'\("(\w+)"', line)
        assert doc == [f[0] for f in getattr(_capi, cls)._fields_], cls


def test_library_choice_of_cooperating_lanes():
    """lgar_cooperating_lanes (host logic, no GPU needed): fp64 jobs under one wave per SIMD get as many lanes per column as
    keep them there -- ceil(N / 1024) columns per wavefront, 64 // that lanes each, at most 16 columns per wavefront -- and
    everything else one lane per column."""
    from lgar_py_amd import _capi
    lib = _capi.load()
    d = _capi.LgarDims()
    d.n_layers, d.num_subcycles, d.nint, d.n_giuh, d.search_mode, d.dt_h = 3, 1, 120, 5, 1, 1.0

    def lanes(n, dtype=_capi.F64, **kw):
        d.n_columns = n
        d.forward_lanes, d.geff_mode, d.use_closed_form_G, d.search_mode, d.nint = 0, 0, 0, 1, 120
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.lgar_cooperating_lanes(C.byref(d), dtype)

    assert [lanes(n) for n in (1, 64, 1024, 1025, 2048, 3000, 10_000, 16_384, 16_385, 1 << 20)] == [64, 64, 64, 32, 32, 21, 6, 4, 1, 1]
    assert lanes(10_000, _capi.F32) == 1                 # fp32: no cooperating kernels
    assert lanes(100, geff_mode=1) == 64                 # mixed-precision trapezoid: its groups of nodes are split as well
    assert lanes(100, use_closed_form_G=1) == 1          # no trapezoid at all
    assert lanes(100, search_mode=0) == 1                # the literal mode
    assert lanes(100, search_mode=2) == 1                # the capacity chain was asked for
    assert lanes(100, nint=200) == 1                     # more intervals than the groups' LDS tables hold
    assert lanes(1 << 20, forward_lanes=8) == 8 and lanes(5, forward_lanes=1) == 1   # the caller's word
    assert lanes(5, forward_lanes=3) < 0                 # rejected (LGAR_E_ARG): a group has at least 4 lanes
