"""Importable alias of the hyphenated package directory ``mri-super-resolution_amd/``.

``import mri_super_resolution_amd`` resolves sub-modules from that directory and executes its
``__init__``; nothing else lives here.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mri-super-resolution_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _os, _f
