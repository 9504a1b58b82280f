"""Shared by tests/test_model_cpu.py and tests/test_gpu_training.py: loss_t + backward against tests/golden/loss_grads.npz."""
import numpy as np
import torch


def check_against_golden_grads(dec, G, tag, dev, tol):
    """loss_t + backward of `dec` on the fixture's inputs (the stored noise replayed) against the reference's numbers."""
    z, mask, mu = (torch.from_numpy(G[tag + "_" + k]).to(dev) for k in ("z", "mask", "mu"))
    t = torch.from_numpy(G[tag + "_t"]).to(dev)
    spk = torch.from_numpy(G[tag + "_spk"]).to(dev) if tag + "_spk" in G else None
    noise = torch.from_numpy(G[tag + "_noise"])
    orig = torch.randn
    torch.randn = lambda *a, **k: noise.to(k.get("device", "cpu"))
    try:
        loss, xt = dec.loss_t(z, mask, mu, t, spk)
    finally:
        torch.randn = orig
    assert float((xt.cpu() - torch.from_numpy(G[tag + "_xt"])).abs().max()) <= 1e-5
    assert abs(float(loss.detach()) - float(G[tag + "_loss"])) <= 2e-5 * abs(float(G[tag + "_loss"]))
    loss.backward()
    grads = {n: p.grad for n, p in dec.estimator.named_parameters() if p.grad is not None}
    names = [str(n) for n in G[tag + "_names"]]
    assert sorted(grads) == sorted(names)
    worst = ("", 0.0)
    for i, n in enumerate(names):
        g = grads[n].detach().double().flatten().cpu().numpy()
        idx = np.unique(np.linspace(0, g.size - 1, 16).round().astype(np.int64))
        scale = float(G[tag + "_max"][i]) + 1e-12
        e = float(np.abs(g[idx] - G[tag + "_vals"][i][:idx.size]).max()) / scale
        e = max(e, abs(float(np.sqrt((g * g).sum())) - float(G[tag + "_norm"][i])) / (float(G[tag + "_norm"][i]) + 1e-12))
        worst = max(worst, (n, e), key=lambda kv: kv[1])
    assert worst[1] <= tol, worst
    return worst


