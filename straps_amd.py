"""Import shim: the package directory is named `straps-3dhumanshapepose_amd/` (not a valid Python
identifier), so `import straps_amd` loads it and aliases it under this name (and under
`straps_3dhumanshapepose_amd`).  Submodules import as `straps_amd.<name>`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, 'straps-3dhumanshapepose_amd')

_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_PKG_DIR, '__init__.py'), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
sys.modules['straps_3dhumanshapepose_amd'] = _mod
_spec.loader.exec_module(_mod)
