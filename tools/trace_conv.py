"""Diagnostic: per-phase s_memtime sums of the 3x3 GroupNorm conv kernel (library built with -DGTTS_TRACE=1)."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = importlib.import_module("speech-backbones_amd")

B, T = int(os.environ.get("TRACE_B", "16")), 1024
dev = torch.device("cuda:0")
def _fixture_state():
    """torch.manual_seed(0) default init of the reference architecture with Rezero.g = 0.02, from the product's own module (as bench.py)."""
    D = importlib.import_module("speech-backbones_amd.model.diffusion")
    torch.manual_seed(0)
    dec = D.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    sd = {k[len("estimator."):]: v.detach().clone() for k, v in dec.state_dict().items()}
    for k in sd:
        if k.endswith(".fn.g"):
            sd[k].fill_(0.02)
    return sd


sd = _fixture_state()
prec = {"bf16x3": S.PREC_BF16X3, "bf16": S.PREC_BF16, "bf16_store": S.PREC_BF16_STORE}[os.environ.get("TRACE_PREC", "bf16x3")]
plan = S.Plan(n_spks=1, precision=prec)
print("precision", os.environ.get("TRACE_PREC", "bf16x3"))
packed = plan.pack(sd, dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 80, T, generator=g).to(dev)
mu = torch.randn(B, 80, T, generator=g).to(dev)
mask = torch.ones(B, 1, T, device=dev)
t = torch.full((B,), 0.5, device=dev)
for _ in range(2):
    out = plan.estimator_forward(packed, x, mask, mu, t)
torch.cuda.synchronize()
lib = S._lib.lib()
buf = (ctypes.c_ulonglong * 3584)()
rc = lib.gtts_debug_trace(buf, 3584)
raw = np.array(buf[:], dtype=np.float64)
a = raw[:2048].reshape(64, 4, 8)
pe = raw[2048:2560].reshape(64, 4, 2)
ep = raw[2560:].reshape(64, 4, 4)
sel = a[:, 0, 6] > 0
a, pe, ep = a[sel], pe[sel], ep[sel]
print("rc", rc, "workgroups traced", a.shape[0], "(last GN 3x3 launch of the call)")
names = ["top barrier", "act transform+write", "weight wait+write", "barrier after w", "prefetch+reads+MFMA", "stage barrier", "loop total", "act load wait"]
NCH = float(os.environ.get('TRACE_CIN', '128')) / 16   # traced layer: cin == cout == TRACE_CIN
tot = a[:, :, 6].mean()
for i, n in enumerate(names):
    print("%-22s mean %10.0f  (%5.1f%% of loop)   per chunk %8.0f" % (n, a[:, :, i].mean(), 100 * a[:, :, i].mean() / tot, a[:, :, i].mean() / NCH))
print("prologue (entry -> loop)   mean %10.0f  (%5.1f%% of loop)" % (pe[:, :, 0].mean(), 100 * pe[:, :, 0].mean() / tot))
print("epilogue (loop -> exit)    mean %10.0f  (%5.1f%% of loop)" % (pe[:, :, 1].mean(), 100 * pe[:, :, 1].mean() / tot))
for i, n in enumerate(["  bias + output stores issued", "  wave reductions + s_red", "  barrier", "  final combine"]):
    print("%-30s mean %10.0f" % (n, ep[:, :, i].mean()))
