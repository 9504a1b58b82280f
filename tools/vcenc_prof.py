#!/usr/bin/env python
"""Diagnostic: DiffVC's average-voice encoder (MelEncoder + PostNet, published sizes) on B x 80 x 1024 mels; run under
rocprofv3 --kernel-trace --stats for the per-kernel table."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VM = importlib.import_module("speech-backbones_amd.diffvc.model")
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = VM.DiffVC(80, 192, 768, 2, 6, 3, 0.1, 4, 128, 128, True, 256, 0.05, 20.0).to(dev).eval()
x = torch.randn(B, 80, 1024, device=dev)
msk = torch.ones(B, 1, 1024, device=dev)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.time()
    m.encoder(x, msk)
    torch.cuda.synchronize()
    print("average-voice encoder B=%d call %d: %.3f ms" % (B, it, (time.time() - t0) * 1e3))
