"""model.monotonic_align -- drop-in for Grad-TTS/model/monotonic_align/__init__.py:8-23.

`maximum_path(value, mask)` keeps the reference's signature and result (0/1 path in value's dtype and on value's
device).  HIP tensors run the DP on the GPU (csrc/mas.hip) instead of copying to the host and looping in Cython: no
device->host->device round trip.  Host tensors -- the reference accepts any device -- run the library's C++ twin
(gtts_mas_maximum_path_cpu).  Both are bit-identical to the reference's core.pyx; neither is a PyTorch fallback.
"""
from .._backend import backend


def maximum_path(value, mask):
    """value, mask: [b, t_x, t_y].  Returns the most likely monotonic alignment (0/1) on value's device."""
    return backend().mas_maximum_path(value, mask)
