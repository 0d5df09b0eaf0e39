"""Import shim: the package directory is `alphazero-rs_amd/` (not a valid Python
identifier), so `import alphazero_rs_amd` loads it from there as a package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "alphazero-rs_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
