"""Regenerate tests/golden/loss_grads.npz from the reference's OWN modules run on CPU: Diffusion.loss_t
(Grad-TTS/model/diffusion.py:281-288) and the gradient of its loss w.r.t. every estimator parameter, single- and multi-speaker.

Run in the build container (where /root/reference is mounted):  python tests/golden/make_golden_grads.py
Weights are re-derived from oracle.gradtts_oracle.make_estimator_state(seed) (checksum kept); the noise draw of
forward_diffusion is stored so that a device with another generator replays it.  Per parameter the file keeps the gradient's
L2 norm, its max |.| and 16 entries at fixed positions -- 172 x 18 numbers instead of 7.6 M."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NS = 16


def sample_index(n):
    return np.unique(np.linspace(0, n - 1, NS).round().astype(np.int64))


def main():
    ref = ref_loader.load_gradtts()
    out = {}
    for tag, n_spks in (("s1", 1), ("s3", 3)):
        seed = 11
        sd = O.make_estimator_state(n_spks=n_spks, seed=seed)
        dec = ref.diffusion.Diffusion(80, 64, n_spks, 64, 0.05, 20.0, 1000)
        dec.estimator.load_state_dict(sd, strict=True)
        inp = O.make_inputs(2, 36, seed=2, spk_dim=64 if n_spks > 1 else None)
        t = torch.tensor([0.35, 0.8])
        torch.manual_seed(4)
        noise = torch.randn(inp["z"].shape)                 # what forward_diffusion draws next (diffusion.py:249)
        torch.manual_seed(4)
        loss, xt = dec.loss_t(inp["z"], inp["mask"], inp["mu"], t, inp.get("spk"))
        loss.backward()
        out[tag + "_seed"] = np.int64(seed)
        out[tag + "_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
        for k in ("z", "mask", "mu"):
            out[tag + "_" + k] = inp[k].numpy()
        if n_spks > 1:
            out[tag + "_spk"] = inp["spk"].numpy()
        out[tag + "_t"] = t.numpy()
        out[tag + "_noise"] = noise.numpy()
        out[tag + "_loss"] = np.float64(float(loss))
        out[tag + "_xt"] = xt.detach().numpy()
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
    np.savez_compressed(os.path.join(OUT, "loss_grads.npz"), **out)


if __name__ == "__main__":
    main()
