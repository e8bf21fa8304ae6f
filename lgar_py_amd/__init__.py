"""Import alias for the package directory `lgar-py_amd/` (a hyphen cannot be imported directly)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "lgar-py_amd")
__path__[:] = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
