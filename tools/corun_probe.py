#!/usr/bin/env python
"""Does an MFMA-bound convolution overlap with a bandwidth-bound kernel of another stream on this chip?
Un-traced: wall time of (a) K convolutions on stream A, (b) K element-wise passes on stream B, (c) both at once.
If (c) ~ max(a, b) the dispatcher co-schedules them; if (c) ~ a + b it does not.   python tools/corun_probe.py [K]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("speech-backbones_amd")
L = pkg._lib
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(0)
x = torch.randn(16, 128, 40, 512, generator=g).to(dev)
w = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).to(dev)
bias = torch.zeros(128, device=dev)
mask = torch.ones(16, 512, device=dev)
a = torch.randn(16, 64, 80, 1024, generator=g).to(dev)
b = torch.randn(16, 64, 80, 1024, generator=g).to(dev)
mask0 = torch.ones(16, 1024, device=dev)


def conv():
    return L.conv3x3_masked(x, mask, w, bias)


def ew():
    return L.add_masked(a, b, mask0)


def run(fa, fb, sa, sb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa is not None:
        with torch.cuda.stream(sa):
            for _ in range(K):
                fa()
    if fb is not None:
        with torch.cuda.stream(sb):
            for _ in range(K):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


for prio in ((0, 0), (0, -1), (-1, 0)):
    sa, sb = torch.cuda.Stream(device=dev, priority=prio[0]), torch.cuda.Stream(device=dev, priority=prio[1])
    for _ in range(2):
        run(conv, ew, sa, sb)
    ta = min(run(conv, None, sa, sb) for _ in range(3))
    tb = min(run(None, ew, sa, sb) for _ in range(3))
    tc = min(run(conv, ew, sa, sb) for _ in range(3))
    t2 = min(run(conv, conv, sa, sb) for _ in range(3))
    t3 = min(run(ew, ew, sa, sb) for _ in range(3))
    print("prio conv/ew %s: conv %.1f us  ew %.1f us  both %.1f us (sum %.1f, max %.1f)  conv+conv %.1f  ew+ew %.1f" %
          (prio, ta * 1e3, tb * 1e3, tc * 1e3, (ta + tb) * 1e3, max(ta, tb) * 1e3, t2 * 1e3, t3 * 1e3), flush=True)

# ---- the same question with a kernel that provably leaves room: the MFMA micro-benchmark at ONE 4-wave workgroup per CU
# (154 VGPRs, no LDS) beside the element-wise pass.  If this pair overlaps and conv || ew does not, the dispatcher is what
# serialises them; if neither overlaps, the power budget is (an MFMA-saturated chip has no headroom for HBM traffic).
import ctypes
lib = L.lib()
rnd = torch.randn(1 << 20, generator=g).to(torch.bfloat16).to(dev)
for wgs in (256, 512):
    sink = torch.empty(int(lib.gtts_ubench_mfma_out_floats(wgs)), dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    iters = 1700 if wgs == 256 else 850

    def mf():
        L._check(lib.gtts_ubench_mfma(L._ptr(rnd), ctypes.c_size_t(rnd.numel() * 2), L._ptr(sink), wgs, iters, ctypes.byref(fl),
                                      L._stream()), "ubench")

    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(2):
        run(mf, ew, sa, sb)
    ta = min(run(mf, None, sa, sb) for _ in range(3))
    tb = min(run(None, ew, sa, sb) for _ in range(3))
    tc = min(run(mf, ew, sa, sb) for _ in range(3))
    print("mfma ubench (%d WGs) %.1f us  ew %.1f us  both %.1f us (sum %.1f, max %.1f)" %
          (wgs, ta * 1e3, tb * 1e3, tc * 1e3, (ta + tb) * 1e3, max(ta, tb) * 1e3), flush=True)
