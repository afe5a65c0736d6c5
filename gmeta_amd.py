"""Import alias: the package directory is named `g-meta_amd/` (not a valid Python identifier), so
`import gmeta_amd` loads it from there.  Submodules resolve normally: `from gmeta_amd.meta import Meta`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g-meta_amd')
_spec = importlib.util.spec_from_file_location('gmeta_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['gmeta_amd'] = _mod
_spec.loader.exec_module(_mod)
