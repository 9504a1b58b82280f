"""Diagnostic (GPU): run-to-run reproducibility of reverse_diffusion by batch size and sub-batch stream count."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
sd = O.make_estimator_state(seed=0)
for streams in (0, 3):
    plan = S.Plan(streams=streams)
    blob = plan.pack(sd, dev)
    for B, T in ((16, 1024), (6, 1024), (5, 1024), (3, 512), (16, 256)):
        inp = O.make_inputs(B, T, seed=1234, ragged=True)
        z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
        outs = [plan.reverse_diffusion(blob, z, m, mu, 2) for _ in range(4)]
        torch.cuda.synchronize()
        nd = [int((outs[0] != o).sum()) for o in outs[1:]]
        mx = [float((outs[0] - o).abs().max()) for o in outs[1:]]
        # which samples differ
        bad = sorted(set(int(i) for o in outs[1:] for i in torch.nonzero((outs[0] != o).flatten(1).any(1)).flatten()))
        print("streams %d B %2d T %4d: differing elements %s max %s samples %s finite %s" % (streams, B, T, nd, mx, bad, bool(torch.isfinite(outs[0]).all())))
