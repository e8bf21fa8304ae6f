"""CPU: the lean fp64 log2 / exp2 / pow of the device code (lgar_py_amd/csrc/lgar_math.hpp, compiled for the host by the
test-only simulator) against mpmath at 40 digits: error bounds stated in that header, special values, and the polynomial
coefficients re-derived."""
import ctypes as C

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")


def _call(op, x, y=None):
    import devsim
    L = devsim.lib(3)
    L.devsim_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.zeros_like(x)
    L.devsim_math(op, len(x), x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return out


@pytest.mark.parametrize("v", [0, 10], ids=["horner", "pairwise"])
def test_exp2_relative_error(v):
    mp.mp.dps = 40
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-60, 60, 4000), rng.uniform(-1, 1, 2000), [0.0, 0.5, -0.5, 1.0, -1022.0, 1023.0]])
    got = _call(v + 0, x)
    ref = [mp.mpf(2) ** mp.mpf(float(u)) for u in x]
    rel = max(abs((mp.mpf(float(g)) - r) / r) for g, r in zip(got, ref))
    assert rel <= 2.5e-16, rel  # one rounding of the result + 3.2e-18 polynomial error
    assert np.array_equal(_call(v + 3, x), got)  # the unclamped core used inside the trapezoid: same values on finite input
    sp = _call(v + 0, [np.nan, -np.inf, np.inf, -2000.0, 2000.0])
    assert np.isnan(sp[0]) and sp[1] == 0.0 and np.isinf(sp[2]) and sp[3] == 0.0 and np.isinf(sp[4])


@pytest.mark.parametrize("v", [0, 10], ids=["horner", "pairwise"])
def test_log2_absolute_error(v):
    mp.mp.dps = 40
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 4000)), rng.uniform(0.5, 2.0, 3000), [1.0, 0.5, 2.0, 5e-324, 1e308]])
    got = _call(v + 1, x)
    err = max(abs(mp.mpf(float(g)) - mp.log(mp.mpf(float(u)), 2)) / max(1, abs(mp.log(mp.mpf(float(u)), 2))) for g, u in zip(got, x))
    assert err <= 2.5e-16, err  # absolute for |log2 x| <= 1, relative beyond
    assert np.array_equal(_call(v + 4, x), got)
    # near x = 1 the result is accurate relative to ITSELF (1 - Se^(1/m) in calc_k_from_se needs (1 - eps)^m - 1)
    eps = np.concatenate([10.0 ** rng.uniform(-17, -2, 2000), -(10.0 ** rng.uniform(-17, -2, 2000))])
    xn = 1.0 + eps
    gn = _call(v + 1, xn)
    nz = xn != 1.0
    reln = max(abs((mp.mpf(float(g)) - mp.log(mp.mpf(float(u)), 2)) / mp.log(mp.mpf(float(u)), 2)) for g, u in zip(gn[nz], xn[nz]))
    assert reln <= 4e-16, reln
    assert (gn[~nz] == 0.0).all()
    sp = _call(v + 1, [0.0, -1.0, np.nan, np.inf])
    assert sp[0] == -np.inf and np.isnan(sp[1]) and np.isnan(sp[2]) and sp[3] == np.inf


@pytest.mark.parametrize("v", [0, 10], ids=["horner", "pairwise"])
def test_pow_in_the_van_genuchten_range(v):
    """pow as the leaf functions use it (physics/utils.py): bases 1e-12 .. 1e6, exponents -12 .. 12."""
    mp.mp.dps = 40
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(np.log(1e-12), np.log(1e6), 6000))
    y = rng.uniform(-12, 12, 6000)
    got = _call(v + 2, x, y)
    worst = 0
    for g, a, b in zip(got, x, y):
        r = mp.mpf(float(a)) ** mp.mpf(float(b))
        if 1e-280 < r < 1e280:
            cond = max(1.0, abs(float(b) * float(mp.log(mp.mpf(float(a)), 2))))
            worst = max(worst, float(abs((mp.mpf(float(g)) - r) / r)) / cond)
    assert worst <= 8e-16, worst  # relative error per unit of max(1, |y log2 x|); observed 6.7e-16
    assert np.isnan(_call(v + 2, [-2.0], [0.5])[0])  # negative base: NaN, what the status word reports as NEGBASE


def test_polynomial_coefficients_are_the_chebyshev_fits():
    mp.mp.dps = 60
    c, err = mp.chebyfit(lambda f: mp.mpf(2) ** f, [-0.5, 0.5], 12, error=True)
    assert float(err) <= 3.2e-18 and abs(float(c[0]) - 4.45581790833606449e-10) <= 1e-24 and abs(float(c[-2]) - 6.93147180559945286e-01) <= 1e-16
    cd = mp.mpf(float(mp.sqrt(mp.mpf(1) / 2)))
    zmax = ((mp.mpf("0.5") - cd) / (mp.mpf("0.5") + cd)) ** 2 * mp.mpf("1.0000001")

    def q(z):
        if z == 0:
            return 2 / mp.log(2)
        s = mp.sqrt(z)
        return 2 / mp.log(2) * mp.atanh(s) / s

    c2, err2 = mp.chebyfit(q, [0, zmax], 8, error=True)
    assert float(err2) <= 3.4e-18 and abs(float(c2[0]) - 2.13658959211262989e-01) <= 1e-15 and abs(float(c2[-1]) - 2.88539008177792677) <= 1e-15
