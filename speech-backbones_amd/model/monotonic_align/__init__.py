"""model.monotonic_align -- drop-in for Grad-TTS/model/monotonic_align/__init__.py:8-23.

`maximum_path(value, mask)` keeps the reference's signature and result (0/1 path in value's dtype) but runs
the DP on the GPU (csrc/mas.hip) instead of copying to the host and looping in Cython: no device->host->device
round trip.  MI355X only: CPU tensors are rejected (there is deliberately no CPU fallback).
"""
from .._backend import backend


def maximum_path(value, mask):
    """value, mask: [b, t_x, t_y] on a HIP device.  Returns the most likely monotonic alignment (0/1)."""
    if not value.is_cuda:
        raise RuntimeError("monotonic_align.maximum_path runs on the MI355X HIP kernel only; got a %s tensor "
                           "(there is no CPU fallback)" % value.device)
    return backend().mas_maximum_path(value, mask)
