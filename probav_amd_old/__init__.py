"""Importable alias of the `proba-v_amd/` package (a hyphen is not a valid Python identifier).

`import probav_amd` executes `proba-v_amd/__init__.py` with this module's `__path__` pointing at that
directory, so `probav_amd.modelsTF`, `probav_amd.loss`, ... resolve to the files under `proba-v_amd/`.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "proba-v_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
