"""Import shim: the package directory is named ``multiple-object-tracking_amd``
(not a valid Python identifier), so it is registered here under the module name
``multiple_object_tracking_amd`` and re-exported as ``mot_amd``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "multiple-object-tracking_amd")
_NAME = "multiple_object_tracking_amd"

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(pkg).items() if not k.startswith("__")})
