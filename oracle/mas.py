"""TEST INFRASTRUCTURE ONLY.  Python faces of the two CPU MAS checkers.

* ``maximum_path_port``  -- the plain-C restatement oracle/mas_oracle.c (always available after `make -C oracle`)
* ``maximum_path_ref``   -- the reference's own Cython kernel compiled by oracle/build_ref.py (oracle/_ref)

Both mirror the wrapper Grad-TTS/model/monotonic_align/__init__.py:8-23: value*mask, float32 numpy,
t_x = mask.sum(1)[:,0], t_y = mask.sum(2)[:,0], int32 path cast back to value's dtype.
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_port = None
_ref = None


def _load_port():
    global _port
    if _port is None:
        so = os.path.join(_HERE, "libmas_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "libmas_oracle.so"], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.mas_oracle_maximum_path.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_float]
        lib.mas_oracle_maximum_path.restype = None
        _port = lib
    return _port


def ref_available():
    from . import build_ref
    return os.path.exists(build_ref.ref_so_path()) or os.path.exists(build_ref.PYX)


def _load_ref():
    global _ref
    if _ref is None:
        from . import build_ref
        so = build_ref.build()
        if so is None or not os.path.exists(so):
            raise RuntimeError("reference MAS not built (oracle/_ref) and /root/reference absent")
        spec = importlib.util.spec_from_file_location("core", so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _ref = mod
    return _ref


def _prep(value, mask):
    value = value * mask
    v = value.data.cpu().numpy().astype(np.float32)
    m = mask.data.cpu().numpy()
    t_x = m.sum(1)[:, 0].astype(np.int32)
    t_y = m.sum(2)[:, 0].astype(np.int32)
    return np.ascontiguousarray(v), np.ascontiguousarray(t_x), np.ascontiguousarray(t_y)


def maximum_path_port(value, mask):
    v, t_x, t_y = _prep(value, mask)
    path = np.zeros(v.shape, dtype=np.int32)
    b, tx, ty = v.shape
    _load_port().mas_oracle_maximum_path(path.ctypes.data, v.ctypes.data, t_x.ctypes.data, t_y.ctypes.data,
                                         b, tx, ty, ctypes.c_float(-1e9))
    return torch.from_numpy(path).to(device=value.device, dtype=value.dtype)


def maximum_path_ref(value, mask):
    v, t_x, t_y = _prep(value, mask)
    path = np.zeros(v.shape, dtype=np.int32)
    _load_ref().maximum_path_c(path, v, t_x, t_y)
    return torch.from_numpy(path).to(device=value.device, dtype=value.dtype)
