"""Locate the ctypes binding whether this package is imported as `speech-backbones_amd.model` or, through
sys.path, as the drop-in top-level `model` package (what Grad-TTS/inference.py imports)."""
import importlib.util
import os
import sys


def backend():
    try:
        from .. import _lib          # imported as speech-backbones_amd.model
        return _lib
    except (ImportError, ValueError):
        name = "gradtts_mi355x_lib"
        if name in sys.modules:
            return sys.modules[name]
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_lib.py")
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
