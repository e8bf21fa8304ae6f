"""ctypes front-end for the CPU oracle (oracle/lgar_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package.  Mirrors the structs of lgar_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LMAX, FMAX, GMAX = 8, 32, 16

ST_NAN, ST_NEGBASE, ST_THETA_ORDER, ST_OVERFLOW, ST_ITERCAP, ST_BOTTOM, ST_STRUCT = 1, 2, 4, 8, 16, 32, 64

ACC_NAMES = ["precip", "PET", "AET", "infiltration", "runoff", "percolation", "giuh_runoff", "discharge",
             "ponded_water", "ending_volume"]


class Front(C.Structure):
    _fields_ = [("depth", C.c_double), ("theta", C.c_double), ("psi", C.c_double), ("k", C.c_double),
                ("dzdt", C.c_double), ("layer", C.c_int), ("to_bottom", C.c_int)]


class Params(C.Structure):
    _fields_ = [("L", C.c_int)] + [(nm, C.c_double * LMAX) for nm in
                                   ("alpha", "n", "m", "ksat", "theta_e", "theta_r", "thick", "cum")] + [
        ("initial_psi", C.c_double), ("pdm", C.c_double), ("wp_psi", C.c_double), ("frozen_factor", C.c_double),
        ("dt_h", C.c_double), ("nint", C.c_int), ("num_subcycles", C.c_int), ("ngiuh", C.c_int),
        ("giuh", C.c_double * GMAX), ("iter_cap", C.c_long), ("closed_form", C.c_int), ("bottom_mode", C.c_int)]


class State(C.Structure):
    _fields_ = [("nf", C.c_int), ("f", Front * FMAX), ("ponded_water", C.c_double), ("previous_precip", C.c_double),
                ("ending_volume", C.c_double), ("giuh_queue", C.c_double * GMAX)] + [
        (nm, C.c_double) for nm in ACC_NAMES[:8]] + [
        ("status", C.c_int), ("n_geff", C.c_long), ("n_tmb_calls", C.c_long), ("n_tmb_iters", C.c_long),
        ("n_ccm_iters", C.c_long)]


_lib = None


def build(force=False):
    """liblgar_oracle.so; with LGAR_ORACLE_SANITIZE=1 in the environment the AddressSanitizer + UBSan build instead (the process
    must then run under LD_PRELOAD of gcc's libasan: tests/test_sanitizers.py)."""
    san = os.environ.get("LGAR_ORACLE_SANITIZE") == "1"
    so = os.path.join(_HERE, "liblgar_oracle_asan.so" if san else "liblgar_oracle.so")
    src = os.path.join(_HERE, "lgar_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if san else []), stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        d, i, p = C.c_double, C.c_int, C.POINTER
        _lib.lgo_theta_from_h.restype = d
        _lib.lgo_theta_from_h.argtypes = [d] * 6 + [p(i)]
        _lib.lgo_se_from_theta.restype = d
        _lib.lgo_se_from_theta.argtypes = [d] * 3
        _lib.lgo_se_from_h.restype = d
        _lib.lgo_se_from_h.argtypes = [d] * 4 + [p(i)]
        _lib.lgo_k_from_se.restype = d
        _lib.lgo_k_from_se.argtypes = [d] * 3 + [p(i)]
        _lib.lgo_h_from_se.restype = d
        _lib.lgo_h_from_se.argtypes = [d] * 4 + [p(i)]
        _lib.lgo_geff.restype = d
        _lib.lgo_geff.argtypes = [d] * 8 + [i, p(i)]
        _lib.lgo_aet.restype = d
        _lib.lgo_aet.argtypes = [d] * 9 + [p(i)]
        _lib.lgo_giuh.restype = d
        _lib.lgo_giuh.argtypes = [p(d), p(d), i, d]
        _lib.lgo_mass_balance.restype = d
        _lib.lgo_mass_balance.argtypes = [p(Params), p(State)]
        _lib.lgo_forward.argtypes = [p(Params), p(State), d, d]
        _lib.lgo_run_columns.argtypes = [i, i, i] + [C.c_void_p] * 6 + [d] * 5 + [i, i, C.c_void_p, i] + [C.c_void_p] * 6 + [i]
    return _lib


def _darr(x, n):
    a = (C.c_double * n)()
    for j, v in enumerate(x):
        a[j] = float(v)
    return a


def make_params(alpha, n, ksat, theta_e, theta_r, thickness, initial_psi=2000.0, pdm=0.0, wp_psi=15495.0,
                frozen_factor=1.0, dt_h=1.0, nint=120, num_subcycles=1, giuh=(0.06, 0.51, 0.28, 0.12, 0.03)):
    p = Params()
    L = len(alpha)
    lib().lgo_params_init(C.byref(p), L, _darr(alpha, L), _darr(n, L), _darr(ksat, L), _darr(theta_e, L),
                          _darr(theta_r, L), _darr(thickness, L), C.c_double(initial_psi), C.c_double(pdm),
                          C.c_double(wp_psi), C.c_double(frozen_factor), C.c_double(dt_h), int(nint),
                          int(num_subcycles), _darr(giuh, len(giuh)), len(giuh))
    return p


def init_state(p):
    s = State()
    lib().lgo_state_init(C.byref(p), C.byref(s))
    return s


def set_fronts(s, fronts, layer, bottom, nf):
    """Inject a front table ([F][5] depth,theta,psi,k,dzdt) into a state."""
    s.nf = int(nf)
    for i in range(int(nf)):
        f = s.f[i]
        f.depth, f.theta, f.psi, f.k, f.dzdt = (float(v) for v in fronts[i])
        f.layer = int(layer[i])
        f.to_bottom = int(bottom[i])


def run(p, s, precip, pet, frec=16, fronts=True):
    """Run T steps with the agent's drain after each; returns dict of per-step arrays."""
    precip = np.ascontiguousarray(precip, dtype=np.float64)
    pet = np.ascontiguousarray(pet, dtype=np.float64)
    T = precip.shape[0]
    acc = np.zeros((T, 10))
    nf = np.zeros((T,), dtype=np.int32)
    fr = np.zeros((T, frec, 5)) if fronts else None
    fl = np.zeros((T, frec), dtype=np.int8) if fronts else None
    fb = np.zeros((T, frec), dtype=np.int8) if fronts else None
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    L = lib()
    L.lgo_run.argtypes = [C.POINTER(Params), C.POINTER(State), C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
    L.lgo_run(C.byref(p), C.byref(s), T, vp(precip), vp(pet), vp(acc), frec, vp(fr), vp(fl), vp(fb), vp(nf))
    return dict(acc=acc, nfronts=nf, fronts=fr, front_layer=fl, front_bottom=fb, status=s.status)


def run_columns(alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, initial_psi=2000.0, pdm=0.0,
                wp_psi=15495.0, frozen_factor=1.0, dt_h=1.0, nint=120, num_subcycles=1,
                giuh=(0.06, 0.51, 0.28, 0.12, 0.03), nthreads=0, want_series=True):
    """Many columns, SoA [L][N] params and [T][N] forcing (fp64).  Returns runoff[T][N], perc[T][N], acc[10][N], status[N]."""
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    alpha, n, ksat, theta_e, theta_r, thickness, precip, pet = map(c, (alpha, n, ksat, theta_e, theta_r, thickness, precip, pet))
    Lc, N = alpha.shape
    T = precip.shape[0]
    ro = np.zeros((T, N)) if want_series else None
    pc = np.zeros((T, N)) if want_series else None
    acc = np.zeros((10, N))
    st = np.zeros((N,), dtype=np.int32)
    g = c(np.array(giuh))
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    lib().lgo_run_columns(N, Lc, T, vp(alpha), vp(n), vp(ksat), vp(theta_e), vp(theta_r), vp(thickness),
                          initial_psi, pdm, wp_psi, frozen_factor, dt_h, nint, num_subcycles, vp(g), len(giuh),
                          vp(precip), vp(pet), vp(ro), vp(pc), vp(acc), vp(st), nthreads)
    return ro, pc, acc, st
