"""Diagnostic (GPU): run the estimator twice with keep_intermediates and report every op output that differs between the
two runs (races), or dump / compare the op outputs of two library builds (GTTS_LIB=... python tools/ws_repro.py dump f.pt).

    python tools/ws_repro.py [B T]                 run-to-run comparison
    python tools/ws_repro.py dump out.pt [B T]     save the op outputs
    python tools/ws_repro.py cmp a.pt b.pt         compare two dumps
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, T):
    from oracle import gradtts_oracle as O
    S = importlib.import_module("speech-backbones_amd")
    dev = torch.device("cuda:0")
    sd = O.make_estimator_state(seed=0)
    inp = O.make_inputs(B, T, seed=1234, ragged=True)
    plan = S.Plan(keep_intermediates=True, streams=0)
    blob = plan.pack(sd, dev)
    t = torch.linspace(0.9, 0.1, B)
    args = (blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev))
    out = plan.estimator_forward(*args)
    torch.cuda.synchronize()
    taps = {k: v.clone() for k, v in plan.tensors(B, T, dev).items()}
    taps["__out__"] = out.clone()
    return plan, args, taps


def main():
    a = sys.argv[1:]
    if a and a[0] == "cmp":
        x, y = torch.load(a[1]), torch.load(a[2])
        for k in x:
            if k in y and x[k].shape == y[k].shape and x[k].dtype == y[k].dtype and x[k].is_floating_point():
                d = float((x[k] - y[k]).abs().max())
                s = float(y[k].abs().max())
                print("%-28s %-22s |max| %10.4g  abs %10.3e  rel %9.2e%s" % (k, tuple(x[k].shape), s, d, d / (s + 1e-30), "  <--" if d / (s + 1e-30) > 1e-4 else ""))
        return
    dump = None
    if a and a[0] == "dump":
        dump = a[1]
        a = a[2:]
    B, T = (int(a[0]), int(a[1])) if len(a) >= 2 else (16, 1024)
    plan, args, t1 = run(B, T)
    if dump:
        keep = {k: v.cpu() for k, v in t1.items() if k.endswith(".raw") or k.endswith(".sc") or k.endswith(".out") or k == "__out__"}
        torch.save(keep, dump)
        print("saved", len(keep), "tensors")
        return
    nbad = 0
    for rep in range(3):
        out = plan.estimator_forward(*args)
        torch.cuda.synchronize()
        t2 = plan.tensors(B, T, torch.device("cuda:0"))
        t2["__out__"] = out
        for k in t1:
            if not torch.equal(t1[k], t2[k]):
                d = (t1[k].float() - t2[k].float()).abs()
                idx = torch.nonzero(d.flatten() > 0)
                print("rep %d DIFF %-28s %s  n=%d  max %.3e  first flat idx %d" % (rep, k, tuple(t1[k].shape), idx.numel(), float(d.max()), int(idx[0])))
                nbad += 1
    print("differing tensors:", nbad)


if __name__ == "__main__":
    main()
