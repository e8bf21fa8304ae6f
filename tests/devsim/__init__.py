"""TEST INFRASTRUCTURE ONLY: ctypes front-end of the device-code simulator (tests/devsim/devsim.cpp).

The simulator is lgar_py_amd/csrc's device code (column physics, per-lane kernel bodies, front-capacity chain) compiled
for the host with -DLGAR_DEVSIM, one lane at a time.  CPU tests use it to run the code the GPU executes against the
reference's golden vectors.  Never imported by the product package; never timed.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_HERE))
CSRC = os.path.join(ROOT, "lgar_py_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FMAX, GMAX, NACC, NSCAL, NCOUNTERS = 32, 8, 10, 11, 4
ACC_NAMES = ["precip", "PET", "AET", "infiltration", "runoff", "percolation", "giuh_runoff", "discharge",
             "ponded_water", "ending_volume"]


class LgarDims(C.Structure):
    _fields_ = [("n_columns", C.c_int32), ("n_layers", C.c_int32), ("n_steps", C.c_int32),
                ("num_subcycles", C.c_int32), ("nint", C.c_int32), ("n_giuh", C.c_int32),
                ("search_mode", C.c_int32), ("bottom_mode", C.c_int32), ("use_closed_form_G", C.c_int32), ("front_slots", C.c_int32),
                ("dt_h", C.c_double), ("initial_psi", C.c_double), ("ponded_depth_max", C.c_double),
                ("wilting_point_psi", C.c_double), ("frozen_factor", C.c_double), ("giuh", C.c_double * GMAX),
                ("iter_cap", C.c_int64), ("forcing_columns", C.c_int32), ("forcing_group", C.c_int32),
                ("tangent_share", C.c_int32), ("geff_mode", C.c_int32), ("forward_lanes", C.c_int32), ("reserved4", C.c_int32)]


class LgarParams(C.Structure):
    _fields_ = [(nm, C.c_void_p) for nm in ("alpha", "n", "ksat", "theta_e", "theta_r", "thickness")]


class LgarState(C.Structure):
    _fields_ = [(nm, C.c_void_p) for nm in ("depth", "theta", "psi", "k", "dzdt", "flags", "n_fronts", "scalars", "totals", "tickets")]


class LgarForcing(C.Structure):
    _fields_ = [("precip", C.c_void_p), ("pet", C.c_void_p)]


class LgarStepOut(C.Structure):
    _fields_ = [("series", C.c_void_p * NACC), ("basin", C.c_void_p), ("weights", C.c_void_p), ("basin_mask", C.c_uint32),
                ("reserved", C.c_uint32), ("counters", C.c_void_p), ("call_sums", C.c_void_p)]


_libs = {}


def lib(n_layers):
    """libdevsim_<L>.so, built on first use (one soil-layer count per library keeps each build under a minute)."""
    if n_layers in _libs:
        return _libs[n_layers]
    # DEVSIM_SANITIZE=1: the AddressSanitizer + UBSan build (the process must run under LD_PRELOAD of clang's asan runtime:
    # tests/test_sanitizers.py)
    san = os.environ.get("DEVSIM_SANITIZE") == "1"
    so = os.path.join(_HERE, "libdevsim_%s%d.so" % ("san_" if san else "", n_layers))
    deps = [os.path.join(_HERE, "devsim.cpp"), os.path.join(ROOT, "include", "lgar.h")] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        if not os.path.exists(CLANG):
            raise RuntimeError("clang++ of the ROCm toolchain not found: cannot build the device-code simulator")
        tmp = "%s.%d.tmp" % (so, os.getpid())  # (several test workers may build the same library at once)
        # (the sanitizer build at -O0: a minute instead of six at -O1; its fixtures run in a second either way)
        extra = ["-O0", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared-libsan"] if san else []
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared"] + extra +
                              ["-I", os.path.join(ROOT, "include"), "-DDEVSIM_LAYERS(X)=X(%d)" % n_layers,
                               os.path.join(_HERE, "devsim.cpp"), "-o", tmp])
        os.replace(tmp, so)
    L = C.CDLL(so)
    p, i32, vp = C.POINTER, C.c_int32, C.c_void_p
    L.devsim_state_init.argtypes = [p(LgarDims), p(LgarParams), p(LgarState), vp, i32]
    L.devsim_forward.argtypes = [p(LgarDims), p(LgarParams), p(LgarState), p(LgarForcing), p(LgarStepOut), vp, i32]
    L.devsim_tangent.argtypes = [p(LgarDims), p(LgarParams), p(LgarParams), p(LgarForcing), vp, vp, vp, vp, vp, i32]
    _libs[n_layers] = L
    return L


def sanitizer_runtime():
    """clang's AddressSanitizer runtime (what a process loading the DEVSIM_SANITIZE build must LD_PRELOAD)."""
    out = subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def prebuild(layers=(2, 3, 4)):
    """Build several layer counts concurrently (conftest calls this once per session)."""
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=len(layers)) as ex:
        list(ex.map(lib, layers))


class SimEngine:
    """Host-memory twin of lgar_py_amd.engine.LgarEngine running the simulator (numpy arrays, same layouts)."""

    def __init__(self, alpha, n, ksat, theta_e, theta_r, thickness, *, n_columns=None, dt_h=1.0, num_subcycles=1,
                 initial_psi=2000.0, ponded_depth_max=0.0, wilting_point_psi=15495.0, frozen_factor=1.0, nint=120,
                 giuh_ordinates=(0.06, 0.51, 0.28, 0.12, 0.03), dtype=np.float64, iter_cap=0, search_mode=1, bottom_mode=0,
                 use_closed_form_G=False, front_slots=None, geff_mode=0):
        self.dtype = np.dtype(dtype)
        self._dt = 1 if self.dtype == np.float64 else 0

        def prep(x):
            t = np.asarray(x, dtype=np.float64)
            if t.ndim == 1:
                t = np.repeat(t[:, None], n_columns, axis=1)
            return np.ascontiguousarray(t.astype(self.dtype))

        self.alpha, self.n, self.ksat = prep(alpha), prep(n), prep(ksat)
        self.theta_e, self.theta_r, self.thickness = prep(theta_e), prep(theta_r), prep(thickness)
        L, N = self.alpha.shape
        self.L, self.N = L, N
        self.lib = lib(L)
        d = self.dims = LgarDims()
        d.n_columns, d.n_layers, d.n_steps, d.num_subcycles = N, L, 0, int(num_subcycles)
        d.nint, d.n_giuh, d.search_mode = int(nint), len(giuh_ordinates), int(search_mode)
        d.dt_h, d.initial_psi, d.ponded_depth_max = float(dt_h), float(initial_psi), float(ponded_depth_max)
        d.wilting_point_psi, d.frozen_factor = float(wilting_point_psi), float(frozen_factor)
        for i, g in enumerate(giuh_ordinates):
            d.giuh[i] = float(g)
        d.iter_cap, d.bottom_mode, d.use_closed_form_G = int(iter_cap), int(bottom_mode), int(bool(use_closed_form_G))
        d.geff_mode = int(geff_mode)
        F = int(front_slots) if front_slots else FMAX
        d.front_slots = F
        z = lambda *s, dt=self.dtype: np.zeros(s, dtype=dt)
        self.depth, self.theta, self.psi, self.k, self.dzdt = z(F, N), z(F, N), z(F, N), z(F, N), z(F, N)
        self.flags, self.n_fronts = z(F, N, dt=np.uint8), z(N, dt=np.int32)
        self.scalars, self.totals, self.status = z(NSCAL, N), z(NACC, N), z(N, dt=np.int32)
        self.counters = z(NCOUNTERS, dt=np.uint64)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._params = LgarParams(*[ptr(t) for t in (self.alpha, self.n, self.ksat, self.theta_e, self.theta_r, self.thickness)])
        self._state = LgarState(*[ptr(t) for t in (self.depth, self.theta, self.psi, self.k, self.dzdt, self.flags,
                                                   self.n_fronts, self.scalars, self.totals)] + [None])
        self.reset()

    def reset(self):
        rc = self.lib.devsim_state_init(C.byref(self.dims), C.byref(self._params), C.byref(self._state),
                                        self.status.ctypes.data_as(C.c_void_p), self._dt)
        assert rc == 0, rc

    def forward(self, precip, pet, series=("runoff", "percolation"), basin=(), weights=None, call_sums=False, forcing_group=1):
        precip = np.ascontiguousarray(np.asarray(precip, dtype=np.float64).astype(self.dtype))
        pet = np.ascontiguousarray(np.asarray(pet, dtype=np.float64).astype(self.dtype))
        T = precip.shape[0]
        assert precip.shape == pet.shape and self.N % (precip.shape[1] * forcing_group) == 0
        self.dims.forcing_columns = precip.shape[1]
        self.dims.forcing_group = forcing_group
        res, so = {}, LgarStepOut()
        for nm in series:
            buf = np.zeros((T, self.N), dtype=self.dtype)
            res[nm] = buf
            so.series[ACC_NAMES.index(nm)] = buf.ctypes.data_as(C.c_void_p)
        keep = []
        if basin:
            block = np.zeros((NACC, T))
            so.basin = block.ctypes.data_as(C.c_void_p)
            for nm in basin:
                so.basin_mask |= 1 << ACC_NAMES.index(nm)
                res["basin:" + nm] = block[ACC_NAMES.index(nm)]
            if weights is not None:
                w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).astype(self.dtype))
                keep.append(w)
                so.weights = w.ctypes.data_as(C.c_void_p)
        so.counters = self.counters.ctypes.data_as(C.c_void_p)
        if call_sums:
            res["call_sums"] = np.zeros((NACC, self.N), dtype=self.dtype)
            so.call_sums = res["call_sums"].ctypes.data_as(C.c_void_p)
        self.dims.n_steps = T
        fo = LgarForcing(precip.ctypes.data_as(C.c_void_p), pet.ctypes.data_as(C.c_void_p))
        rc = self.lib.devsim_forward(C.byref(self.dims), C.byref(self._params), C.byref(self._state), C.byref(fo), C.byref(so),
                                     self.status.ctypes.data_as(C.c_void_p), self._dt)
        assert rc == 0, rc
        return res

    def tangent(self, direction, precip, pet, w_runoff=None, w_perc=None, want_series=False, forcing_group=1):
        prep = lambda t: None if t is None else np.ascontiguousarray(np.asarray(t, dtype=np.float64).astype(self.dtype))
        precip, pet, w_runoff, w_perc = prep(precip), prep(pet), prep(w_runoff), prep(w_perc)
        T = precip.shape[0]
        assert self.N % (precip.shape[1] * forcing_group) == 0
        self.dims.forcing_columns = precip.shape[1]
        self.dims.forcing_group = forcing_group
        dirs = {k: prep(direction.get(k)) for k in ("alpha", "n", "ksat")}
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        dstruct = LgarParams(ptr(dirs["alpha"]), ptr(dirs["n"]), ptr(dirs["ksat"]), None, None, None)
        grad = np.zeros(self.N, dtype=self.dtype)
        ser = np.zeros((T, self.N), dtype=self.dtype) if want_series else None
        st = np.zeros(self.N, dtype=np.int32)
        self.dims.n_steps = T
        fo = LgarForcing(ptr(precip), ptr(pet))
        rc = self.lib.devsim_tangent(C.byref(self.dims), C.byref(self._params), C.byref(dstruct), C.byref(fo), ptr(w_runoff),
                                     ptr(w_perc), ptr(grad), ptr(ser), ptr(st), self._dt)
        assert rc == 0, rc
        return grad, ser, st

    def fronts(self):
        fl = self.flags
        return dict(depth=self.depth, theta=self.theta, psi=self.psi, k=self.k, dzdt=self.dzdt,
                    layer=(fl & 0x7F).astype("int8"), to_bottom=(fl >> 7).astype("int8"), n_fronts=self.n_fronts)
