"""Import shim: the package lives in ``random-envs_amd/`` (a directory name Python cannot
import directly because of the hyphen); this module makes it importable as ``random_envs_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "random-envs_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real
