"""Import shim: the package directory is named ``whisper.axera_amd`` (with a dot), which the
import statement cannot spell. ``import whisper_axera_amd`` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "whisper.axera_amd")
_spec = importlib.util.spec_from_file_location("whisper_axera_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["whisper_axera_amd"] = _mod
_spec.loader.exec_module(_mod)
