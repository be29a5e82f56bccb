'''
Import alias: ``import rcf_amd`` == the package in ./radar-camera-fusion-depth_amd/ (whose prescribed directory name
is not a Python identifier).  Every submodule is registered under both names so ``rcf_amd.x`` and
``radar-camera-fusion-depth_amd.x`` are the SAME module object (one copy of the loaded library handle, one class
identity).
'''
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = 'radar-camera-fusion-depth_amd'
_pkg = importlib.import_module(_REAL)
for _f in sorted(os.listdir(os.path.join(_root, _REAL))):
    if _f.endswith('.py') and _f != '__init__.py':
        _name = _f[:-3]
        _mod = importlib.import_module(_REAL + '.' + _name)
        sys.modules[__name__ + '.' + _name] = _mod
        setattr(_pkg, _name, _mod)
sys.modules[__name__] = _pkg
