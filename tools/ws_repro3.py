"""Diagnostic (GPU): two estimator calls running concurrently on two streams (own plans / workspaces) vs the same calls
run alone; prints the op outputs that differ."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
sd = O.make_estimator_state(seed=0)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 1024)
plans = [S.Plan(keep_intermediates=True, streams=0) for _ in range(2)]
blobs = [p.pack(sd, dev) for p in plans]
ins = []
for i in range(2):
    inp = O.make_inputs(B, T, seed=1234 + i, ragged=True)
    ins.append((inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), torch.linspace(0.9, 0.1, B).to(dev)))
ref = []
for i in range(2):
    out = plans[i].estimator_forward(blobs[i], *ins[i])
    torch.cuda.synchronize()
    d = {k: v.clone() for k, v in plans[i].tensors(B, T, dev).items()}
    d["__out__"] = out.clone()
    ref.append(d)
streams = [torch.cuda.Stream() for _ in range(2)]
for rep in range(4):
    outs = [None, None]
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs[i] = plans[i].estimator_forward(blobs[i], *ins[i])
    torch.cuda.synchronize()
    for i in range(2):
        cur = plans[i].tensors(B, T, dev)
        cur["__out__"] = outs[i]
        names = list(ref[i].keys())
        bad = [(k, cur[k]) for k in names if not torch.equal(ref[i][k], cur[k])]
        for k, v in bad[:6]:
            d = (ref[i][k].float() - v.float()).abs()
            nz = torch.nonzero(d.flatten() > 0).flatten()
            bs = sorted(set(int(x) for x in (nz // (d.numel() // d.shape[0])).tolist()))[:8] if d.dim() > 1 else []
            print("rep %d plan %d DIFF %-26s %-22s n=%d max %.3e samples %s first %d" % (rep, i, k, tuple(v.shape), nz.numel(), float(d.max()), bs, int(nz[0])))
        print("rep %d plan %d: %d differing tensors" % (rep, i, len(bad)))
