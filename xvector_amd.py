"""Import shim: the product package lives in `speaker-recognition-x-vectors_amd/`
(a directory name that is not a Python identifier).  `import xvector_amd` loads that
package under this importable alias."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "speaker-recognition-x-vectors_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
