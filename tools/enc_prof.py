#!/usr/bin/env python
"""Diagnostic: the Grad-TTS text encoder (180 tokens) at B = 1 / 16 through the drop-in module; run under
rocprofv3 --kernel-trace --stats for the per-kernel table."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("speech-backbones_amd.model")
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.GradTTS(149, 1, 64, 192, 768, 256, 2, 6, 3, 0.1, 4, 80, 64, 0.05, 20.0, 1000).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randint(1, 149, (B, 180)).to(dev)
xl = torch.full((B,), 180, dtype=torch.long, device=dev)
with torch.no_grad():
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.time()
        model.encoder(x, xl)
        torch.cuda.synchronize()
        print("encoder B=%d call %d: %.3f ms" % (B, it, (time.time() - t0) * 1e3))
