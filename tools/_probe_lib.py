"""The probe tools run against libgmeta_hip_probes.so (the library's sources with -DGM_PROBES: include/gmeta_hip_probes.h), never against the
product library: importing this module builds it when it is missing and points GMETA_HIP_LIB at it -- import it BEFORE gmeta_amd."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location('gmeta_amd_build', os.path.join(ROOT, 'g-meta_amd', 'build.py'))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
os.environ['GMETA_HIP_LIB'] = _mod.build(probes=True)
