"""Diagnostic (GPU): the same U-Net call on random data / weights and on all-zero data / weights.  Same instruction stream,
different switching activity: the ratio shows how much of the time is the power-managed clock rather than the kernels' issue."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
B, T = 16, 1024
sd = O.make_estimator_state(seed=0)
sd0 = {k: torch.zeros_like(v) for k, v in sd.items()}
inp = O.make_inputs(B, T, seed=1234, ragged=False)
t = torch.full((B,), 0.5)
for name, w, scale in (("random", sd, 1.0), ("zeros", sd0, 0.0), ("random", sd, 1.0), ("zeros", sd0, 0.0)):
    plan = S.Plan(streams=0)
    blob = plan.pack(w, dev)
    args = (blob, (inp["z"] * scale).to(dev), inp["mask"].to(dev), (inp["mu"] * scale).to(dev), t.to(dev))
    for _ in range(3):
        plan.estimator_forward(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.estimator_forward(*args)
    e1.record()
    torch.cuda.synchronize()
    print("%-7s %.3f ms per U-Net call" % (name, e0.elapsed_time(e1) / 20))
