"""Import alias: ``import vidmem`` loads the package that lives in ``real-time-brain-inspired-video-memory_amd/``.

The package directory carries the repository's mandated name, which is not a valid Python identifier (hyphens);
this module replaces itself in ``sys.modules`` with that package so ``vidmem.memory`` etc. resolve normally.
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real-time-brain-inspired-video-memory_amd")
_spec = importlib.util.spec_from_file_location(
    "vidmem", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vidmem"] = _mod
_spec.loader.exec_module(_mod)
