"""ctypes binding of the C-ABI declared in include/lgar.h (liblgar_hip.so).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, every entry point
raises.  Torch is used only for device memory and streams; the signatures carry raw pointers.
"""
import ctypes as C
import os

from . import build as _build

FMAX, LMIN, LMAX, GMAX = 32, 2, 6, 8
CAP_SMALL, CAP_MID = 8, 16
NCOUNTERS = 4
NTICKETS = 8
ST_RESUME, ST_FAULT_MASK = 128, 0x7F
NSCAL = 3 + GMAX
NACC = 10
F32, F64 = 0, 1
ACC_NAMES = ["precip", "PET", "AET", "infiltration", "runoff", "percolation", "giuh_runoff", "discharge",
             "ponded_water", "ending_volume"]
ST_NAN, ST_NEGBASE, ST_THETA_ORDER, ST_OVERFLOW, ST_ITERCAP, ST_BOTTOM, ST_STRUCT = 1, 2, 4, 8, 16, 32, 64
STATUS_NAMES = {1: "NaN", 2: "negative pow base", 4: "theta order", 8: "front overflow", 16: "iteration cap",
                32: "front reached domain bottom", 64: "structural error"}
ABI_VERSION = 3  # LGAR_ABI_VERSION of include/lgar.h this binding mirrors
EXPORTS = ["lgar_version", "lgar_abi_version", "lgar_sizeof_dims", "lgar_fmax", "lgar_lmax", "lgar_cooperating_lanes", "lgar_state_init", "lgar_forward", "lgar_forward_tangent",
           "lgar_leaf_batch", "lgar_valu_probe", "lgar_valu_probe_insts"]


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
    _fields_ = [(nm, C.c_void_p) for nm in ("depth", "theta", "psi", "k", "dzdt", "flags", "n_fronts", "scalars",
                                            "totals", "tickets")]


class LgarForcing(C.Structure):
    _fields_ = [("precip", C.c_void_p), ("pet", C.c_void_p)]


class LgarStepOut(C.Structure):
    _fields_ = [("series", C.c_void_p * NACC), ("basin", C.c_void_p), ("weights", C.c_void_p), ("basin_mask", C.c_uint32), ("reserved", C.c_uint32), ("counters", C.c_void_p), ("call_sums", C.c_void_p)]


class LgarError(RuntimeError):
    pass


_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load liblgar_hip.so, (re)building it first when it is missing or was built from other sources (content
    fingerprint, build.py); raise loudly if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.environ.get("LGAR_LIB"):  # an explicitly selected library (measurement variants) is taken as it is
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise LgarError("liblgar_hip.so is missing or stale and could not be built (%s); there is no CPU fallback" % e)
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise LgarError("cannot load %s: %s; there is no CPU fallback" % (path, e))
    p = C.POINTER
    i32, vp, dbl = C.c_int32, C.c_void_p, C.c_double
    # an older library (a measurement variant selected with LGAR_LIB, a stale copy) would misread LgarDims or take garbage for
    # an argument it does not know: refuse it before the first call
    if not hasattr(lib, "lgar_abi_version"):
        raise LgarError("%s predates lgar_abi_version(): rebuild it (ABI %d expected)" % (path, ABI_VERSION))
    lib.lgar_abi_version.restype = i32
    lib.lgar_sizeof_dims.restype = i32
    if lib.lgar_abi_version() != ABI_VERSION or lib.lgar_sizeof_dims() != C.sizeof(LgarDims):
        raise LgarError("%s has ABI %d / sizeof(LgarDims) %d, this binding needs %d / %d: rebuild it"
                        % (path, lib.lgar_abi_version(), lib.lgar_sizeof_dims(), ABI_VERSION, C.sizeof(LgarDims)))
    lib.lgar_version.restype = C.c_char_p
    lib.lgar_fmax.restype = i32
    lib.lgar_lmax.restype = i32
    lib.lgar_cooperating_lanes.restype = i32
    lib.lgar_cooperating_lanes.argtypes = [p(LgarDims), i32]
    lib.lgar_state_init.restype = i32
    lib.lgar_state_init.argtypes = [p(LgarDims), p(LgarParams), p(LgarState), vp, i32, vp]
    lib.lgar_forward.restype = i32
    lib.lgar_forward.argtypes = [p(LgarDims), p(LgarParams), p(LgarState), p(LgarForcing), p(LgarStepOut), vp, i32, vp]
    if hasattr(lib, "lgar_forward_tangent") or not os.environ.get("LGAR_LIB"):  # measurement variants omit it
        lib.lgar_forward_tangent.restype = i32
        lib.lgar_forward_tangent.argtypes = [p(LgarDims), p(LgarParams), p(LgarParams), p(LgarForcing), vp, vp, vp, vp, vp,
                                             i32, vp, vp]
    lib.lgar_leaf_batch.restype = i32
    lib.lgar_leaf_batch.argtypes = [i32, i32, vp, vp, dbl, vp, vp, vp, vp, vp, i32, dbl, vp, i32, vp]
    lib.lgar_valu_probe_insts.restype = i32
    lib.lgar_valu_probe_insts.argtypes = [i32]
    lib.lgar_valu_probe.restype = i32
    lib.lgar_valu_probe.argtypes = [i32, i32, i32, i32, vp, vp]
    if lib.lgar_fmax() != FMAX or lib.lgar_lmax() != LMAX:
        raise LgarError("liblgar_hip.so was built with different LGAR_FMAX/LGAR_LMAX than the Python binding")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise LgarError("%s failed with code %d (%s)" % (what, rc, {-1: "bad argument", -2: "launch error",
                                                                     -3: "no device"}.get(rc, "?")))
