"""Import alias: ``hybrid-drt_amd/`` (the package directory the repo layout prescribes) is not a valid
Python identifier, so ``import hipdrt`` resolves to it.  Nothing else lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hybrid-drt_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
