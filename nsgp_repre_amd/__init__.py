"""Import alias: the product package lives in ``nsgp-repre_amd/`` (a directory
name Python cannot import because of the hyphen).  This shim points the
importable name ``nsgp_repre_amd`` at that directory and runs its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nsgp-repre_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
