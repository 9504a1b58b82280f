"""Regenerate tests/golden/vc_loss_grads.npz from the reference's OWN DiffVC modules run on CPU: Diffusion.loss_t
(DiffVC/model/diffusion.py:207-218) and the gradient of its loss w.r.t. every decoder parameter, at dim_base 64 and 256.

Run in the build container (where /root/reference is mounted):  python tests/golden/make_golden_grads_vc.py
Weights are re-derived from oracle.diffvc_oracle.make_state(seed) (checksum kept); the noise draw of forward_diffusion is stored
so that a device with another generator replays it.  Per parameter the file keeps the gradient's L2 norm, its max |.| and 16
entries at fixed positions.  (Conv biases in front of an InstanceNorm have an identically-zero gradient -- 1e-10 of rounding noise
in the reference: they are stored with their tiny max and compared on the scale of their weight's gradient.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import diffvc_oracle as V  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NS = 16
CASES = (("d64", 64, 2, 32), ("d256", 256, 1, 16))        # tag, dim_base, B, T


def sample_index(n):
    return np.unique(np.linspace(0, n - 1, NS).round().astype(np.int64))


def make_inputs(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    mask = torch.ones(B, 1, T)
    if B > 1:
        mask[1, :, (3 * T) // 4:] = 0
    x0, mean, xref, mref = (torch.randn(B, 80, T, generator=g) * mask for _ in range(4))
    c = torch.randn(B, 256, generator=g) * 0.3
    t = torch.linspace(0.3, 0.7, B)
    return {"x0": x0, "mask": mask, "mean": mean, "x_ref": xref, "mean_ref": mref, "c": c, "t": t}


def main():
    ref = ref_loader.load_diffvc()
    out = {}
    for tag, dim, B, T in CASES:
        seed = 21
        sd = V.make_state(dim_base=dim, dim_cond=128, use_ref_t=True, seed=seed, rezero_g=0.3)
        dec = ref.diffusion.Diffusion(80, dim, 128, True, 0.05, 20.0)
        dec.estimator.load_state_dict(sd, strict=True)
        inp = make_inputs(B, T, seed=5)
        torch.manual_seed(4)
        noise = torch.randn(inp["x0"].shape)            # what forward_diffusion draws next (DiffVC/model/diffusion.py:159)
        torch.manual_seed(4)
        loss = dec.loss_t(inp["x0"], inp["mask"], inp["mean"], inp["x_ref"], inp["mean_ref"], inp["c"], inp["t"])
        loss.backward()
        out[tag + "_seed"] = np.int64(seed)
        out[tag + "_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
        for k, v in inp.items():
            out[tag + "_" + k] = v.numpy()
        out[tag + "_noise"] = noise.numpy()
        out[tag + "_loss"] = np.float64(float(loss))
        names, norms, maxs, vals = [], [], [], []
        for name, p in dec.estimator.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().double().flatten().numpy()
            idx = sample_index(g.size)
            v = np.zeros(NS)
            v[:idx.size] = g[idx]
            names.append(name)
            norms.append(np.sqrt((g * g).sum()))
            maxs.append(np.abs(g).max())
            vals.append(v)
        out[tag + "_names"] = np.array(names)
        out[tag + "_norm"] = np.array(norms)
        out[tag + "_max"] = np.array(maxs)
        out[tag + "_vals"] = np.stack(vals)
        print(tag, "loss", float(loss), "parameters with a gradient", len(names))
    np.savez_compressed(os.path.join(OUT, "vc_loss_grads.npz"), **out)


if __name__ == "__main__":
    main()
