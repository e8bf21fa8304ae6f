"""lgar_py_amd: MI355X-native many-column LGAR infiltration engine (hot path of LGAR-py / dpLGAR).

(`lgar-py_amd` at the repo root is a symlink to this directory: a hyphen cannot be imported.)
"""
from ._capi import ACC_NAMES, FMAX, LMAX, LgarError  # noqa: F401
from .engine import LgarEngine, LgarStatusError, leaf_batch  # noqa: F401

__all__ = ["LgarEngine", "LgarError", "LgarStatusError", "leaf_batch", "ACC_NAMES", "FMAX", "LMAX"]
