"""Regenerate tests/golden/*.npz from the reference's OWN Python modules run on CPU.

Run in the build container (where /root/reference is mounted):  python tests/golden/make_golden.py
The reference cannot travel to the GPU box, so its outputs are committed as small fixtures.  Weights
are not stored: they are re-derived from oracle.gradtts_oracle.make_estimator_state(seed) and guarded
by a checksum kept in each file.

What each file pins (reference symbol -> arrays):
  est_1spk.npz   GradLogPEstimator2d.forward        diffusion.py:174-216   B=2,T=64, ragged mask
  est_3ch.npz    same, n_spks=4 (3-channel input + spk_mlp)  diffusion.py:139-141,183-185
  rd_ode.npz     Diffusion.reverse_diffusion stoc=False N=4  diffusion.py:254-275
  rd_sde.npz     Diffusion.reverse_diffusion stoc=True  N=3, per-step randn injected
  mas.npz        monotonic_align.maximum_path (compiled core.pyx) on ragged random values
  utils.npz      sequence_mask / fix_len_compatibility / generate_path   model/utils.py:6-39
  vc_dim64.npz   DiffVC GradLogPEstimator.forward + Diffusion.forward ('pf','em','ml', N=3, injected noise)
                 DiffVC/model/diffusion.py:61-106,164-205 with dim_unet=64, use_ref_t=True (RefBlock path)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O  # noqa: E402
from oracle import mas as MAS  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def np_(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    torch.set_num_threads(1)
    ref = ref_loader.load_gradtts()
    RD = ref.diffusion

    # ---- single speaker estimator + reverse diffusion
    sd = O.make_estimator_state(seed=0)
    dec = RD.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.estimator.load_state_dict(sd, strict=True)
    inp = O.make_inputs(2, 64, seed=1234)
    t = torch.tensor([0.7, 0.31])
    with torch.no_grad():
        est = dec.estimator(inp["z"], inp["mask"], inp["mu"], t)
        rd4 = dec(inp["z"], inp["mask"], inp["mu"], 4)
    np.savez_compressed(os.path.join(OUT, "est_1spk.npz"), seed=0, wsum=checksum(sd), t=t.numpy(),
                        **np_({k: inp[k] for k in ("z", "mu", "mask")}), est=est.numpy())
    np.savez_compressed(os.path.join(OUT, "rd_ode.npz"), seed=0, wsum=checksum(sd), n=4,
                        **np_({k: inp[k] for k in ("z", "mu", "mask")}), out=rd4.numpy())

    # ---- SDE branch with injected noise (the reference draws torch.randn per step, diffusion.py:267)
    g = torch.Generator().manual_seed(99)
    noise = torch.randn(3, 2, 80, 64, generator=g)
    calls = {"i": 0}
    real_randn = torch.randn

    def fake_randn(*a, **k):
        i = calls["i"]
        calls["i"] += 1
        return noise[i].clone()

    torch.randn = fake_randn
    try:
        with torch.no_grad():
            rd3 = dec(inp["z"], inp["mask"], inp["mu"], 3, stoc=True)
    finally:
        torch.randn = real_randn
    assert calls["i"] == 3
    np.savez_compressed(os.path.join(OUT, "rd_sde.npz"), seed=0, wsum=checksum(sd), n=3, noise=noise.numpy(),
                        **np_({k: inp[k] for k in ("z", "mu", "mask")}), out=rd3.numpy())

    # ---- multi-speaker (3 input channels, spk_mlp)
    sd3 = O.make_estimator_state(seed=3, n_spks=4)
    dec3 = RD.Diffusion(80, 64, 4, 64, 0.05, 20.0, 1000)
    dec3.estimator.load_state_dict(sd3, strict=True)
    inp3 = O.make_inputs(2, 32, seed=77, spk_dim=64)
    t3 = torch.tensor([0.9, 0.9])
    with torch.no_grad():
        est3 = dec3.estimator(inp3["z"], inp3["mask"], inp3["mu"], t3, inp3["spk"])
    np.savez_compressed(os.path.join(OUT, "est_3ch.npz"), seed=3, wsum=checksum(sd3), t=t3.numpy(),
                        **np_({k: inp3[k] for k in ("z", "mu", "mask", "spk")}), est=est3.numpy())

    # ---- MAS (compiled reference core.pyx)
    g = torch.Generator().manual_seed(5)
    b, tx, ty = 5, 24, 56
    value = torch.randn(b, tx, ty, generator=g) * 3
    xl = torch.tensor([24, 17, 9, 1, 24])
    yl = torch.tensor([56, 40, 9, 30, 24])
    mask = (O.sequence_mask(xl, tx).unsqueeze(-1) * O.sequence_mask(yl, ty).unsqueeze(1)).float()
    path = MAS.maximum_path_ref(value, mask)
    np.savez_compressed(os.path.join(OUT, "mas.npz"), value=value.numpy(), mask=mask.numpy().astype(np.uint8),
                        path=path.numpy().astype(np.uint8))

    # ---- model/utils.py
    U = ref.utils
    lens = torch.tensor([5, 1, 8])
    sm = U.sequence_mask(lens, 9)
    fl = np.array([U.fix_len_compatibility(n) for n in range(0, 20)])
    dur = torch.tensor([[2., 0., 3., 1.], [1., 1., 1., 0.]])
    pm = (U.sequence_mask(torch.tensor([4, 3]), 4).unsqueeze(-1) *
          U.sequence_mask(torch.tensor([6, 3]), 8).unsqueeze(1)).float()
    gp = U.generate_path(dur, pm)
    np.savez_compressed(os.path.join(OUT, "utils.npz"), lens=lens.numpy(), seqmask=sm.numpy(), fixlen=fl,
                        dur=dur.numpy(), pmask=pm.numpy(), path=gp.numpy())
    # ---- DiffVC decoder (DiffVC/model/diffusion.py:17-205), dim_unet=64 keeps the fixture small
    from oracle import diffvc_oracle as V
    vc = ref_loader.load_diffvc()
    sdv = V.make_state(dim_base=64, dim_cond=128, use_ref_t=True, seed=0)
    dvc = vc.diffusion.Diffusion(80, 64, 128, True, 0.05, 20.0)
    dvc.estimator.load_state_dict(sdv, strict=True)
    iv = V.make_inputs(2, 32, 24, seed=7)
    tv = torch.tensor([0.7, 0.3])
    xt_ref = torch.stack([V.compute_diffused_mean(iv["ref"], iv["ref_mask"], iv["mean_ref"], 0.7)], 1)
    with torch.no_grad():
        estv = dvc.estimator(iv["z"], iv["mask"], iv["mean"], xt_ref, iv["ref_mask"], iv["c"], tv)
    g = torch.Generator().manual_seed(3)
    noise_v = torch.randn(3, 2, 80, 32, generator=g)
    outs = {}
    real_rl = torch.randn_like
    for mode in ("pf", "em", "ml"):
        calls = {"i": 0}

        def fake_rl(x, **k):
            i = calls["i"]
            calls["i"] += 1
            return noise_v[i].clone()

        torch.randn_like = fake_rl
        try:
            with torch.no_grad():
                outs[mode] = dvc(iv["z"], iv["mask"], iv["mean"], iv["ref"], iv["ref_mask"], iv["mean_ref"], iv["c"], 3,
                                 mode).numpy()
        finally:
            torch.randn_like = real_rl
    np.savez_compressed(os.path.join(OUT, "vc_dim64.npz"), seed=0, wsum=checksum(sdv), t=tv.numpy(), xt_ref=xt_ref.numpy(),
                        noise=noise_v.numpy(), est=estv.numpy(), out_pf=outs["pf"], out_em=outs["em"], out_ml=outs["ml"],
                        **np_(iv))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
