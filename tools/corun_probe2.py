#!/usr/bin/env python
"""Why do an MFMA-saturating kernel and an HBM-streaming kernel of two streams not overlap on MI355X?  (tools/corun_probe.py
found t(both) ~ t(a) + t(b) even when the MFMA kernel leaves 7 of 8 wave slots per SIMD free.)  Discriminator: the same MFMA
stream on ZERO operands draws far less power (2.43 vs 1.80 PFLOP/s alone).  If zero-operand MFMAs DO overlap with the HBM stream,
the power budget is what serialises the random-operand pair; per-stream event times show who waits for whom."""
import ctypes
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("speech-backbones_amd")
L = pkg._lib
lib = L.lib()
dev = torch.device("cuda:0")
K = 40
g = torch.Generator().manual_seed(0)
rnd = torch.randn(1 << 20, generator=g).to(torch.bfloat16).to(dev)
zero = torch.zeros(1 << 20, dtype=torch.bfloat16, device=dev)
n = 1 << 27
a = torch.randn(n, generator=g).to(dev)
b = torch.randn(n, generator=g).to(dev)
c = torch.empty(n, device=dev)
wgs = 256
sink = torch.empty(int(lib.gtts_ubench_mfma_out_floats(512)), dtype=torch.float32, device=dev)
fl = ctypes.c_double(0.0)
nb = ctypes.c_double(0.0)


def mk_mfma(src, iters, w=wgs):
    def f():
        L._check(lib.gtts_ubench_mfma(L._ptr(src), ctypes.c_size_t(src.numel() * 2), L._ptr(sink), w, iters, ctypes.byref(fl), L._stream()), "mfma")
    return f


def mk_hbm(mode):
    def f():
        L._check(lib.gtts_ubench_hbm(L._ptr(a), L._ptr(b), L._ptr(c), ctypes.c_size_t(n), mode, 256 * 8, ctypes.byref(nb), L._stream()), "hbm")
    return f


def run(fa, fb):
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa is not None:
        with torch.cuda.stream(sa):
            ev[0].record()
            for _ in range(K):
                fa()
            ev[1].record()
    if fb is not None:
        with torch.cuda.stream(sb):
            ev[2].record()
            for _ in range(K):
                fb()
            ev[3].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / K
    ta = ev[0].elapsed_time(ev[1]) * 1e3 / K if fa is not None else 0.0
    tb = ev[2].elapsed_time(ev[3]) * 1e3 / K if fb is not None else 0.0
    return wall, ta, tb


def pair(name, fa, fb):
    for _ in range(2):
        run(fa, fb)
    wa = min(run(fa, None) for _ in range(3))
    wb = min(run(None, fb) for _ in range(3))
    wc = min(run(fa, fb) for _ in range(3))
    print("%-44s a alone %.1f us   b alone %.1f us   both: wall %.1f us (a's stream %.1f, b's stream %.1f)   sum %.1f  max %.1f" %
          (name, wa[0], wb[0], wc[0], wc[1], wc[2], wa[0] + wb[0], max(wa[0], wb[0])), flush=True)


# iters tuned so that each launch is ~250 us alone
pair("mfma random (1 wave/SIMD) || triad", mk_mfma(rnd, 1700), mk_hbm(1))
pair("mfma ZERO operands (1 wave/SIMD) || triad", mk_mfma(zero, 2400), mk_hbm(1))
pair("mfma random || read-only sweep", mk_mfma(rnd, 1700), mk_hbm(2))
pair("mfma random (2 waves/SIMD) || triad", mk_mfma(rnd, 850, 512), mk_hbm(1))
pair("mfma random half rate (128 WGs) || triad", mk_mfma(rnd, 1700, 128), mk_hbm(1))
pair("triad || triad", mk_hbm(1), mk_hbm(1))
