"""lgar-py_amd: MI355X-native many-column LGAR infiltration engine (hot path of LGAR-py / dpLGAR).

The directory name is not a valid Python identifier; import it as `lgar_py_amd` (the sibling alias
package at the repo root points its __path__ here).
"""
from ._capi import ACC_NAMES, FMAX, LMAX, LgarError  # noqa: F401
from .engine import LgarEngine, LgarStatusError, leaf_batch  # noqa: F401

__all__ = ["LgarEngine", "LgarError", "LgarStatusError", "leaf_batch", "ACC_NAMES", "FMAX", "LMAX"]
