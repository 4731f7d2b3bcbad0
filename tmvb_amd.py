"""
tmvb_amd -- import shim for the product package.

The package directory is named `topicmodelsvb.jl_amd/` (after the reference repository); a dot is
not legal in a Python module name, so this shim loads it under the importable name `tmvb_amd_pkg`
and re-exports it:   import tmvb_amd; tmvb_amd.pkg.gpuLDA(...)   or   from tmvb_amd import *.
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "topicmodelsvb.jl_amd")
_NAME = "tmvb_amd_pkg"

if _NAME in sys.modules:
    pkg = sys.modules[_NAME]
else:
    _spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG_DIR, "__init__.py"),
                                                   submodule_search_locations=[_PKG_DIR])
    pkg = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = pkg
    _spec.loader.exec_module(pkg)

globals().update({k: getattr(pkg, k) for k in getattr(pkg, "__all__", [])})
