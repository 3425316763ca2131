"""Import shim: the package directory is named ``video-learning-tf_amd`` (not a Python identifier),
so ``import vltf_amd`` loads it under this name.  Nothing else lives here."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-learning-tf_amd")
_spec = importlib.util.spec_from_file_location("vltf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vltf_amd"] = _mod
_spec.loader.exec_module(_mod)
