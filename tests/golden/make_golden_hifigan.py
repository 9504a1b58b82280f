"""Golden vectors of the HiFi-GAN generator, produced by the REFERENCE's own module (Grad-TTS/hifi-gan/models.py)
in the container where /root/reference is mounted:   python tests/golden/make_golden_hifigan.py
Weights come from oracle.hifigan_oracle.make_state(seed) (deterministic, not stored); inputs and outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hifigan_oracle as H  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run_reference(ref, cfg, sd, mel):
    h = ref.AttrDict(dict(cfg, upsample_rates=list(cfg["upsample_rates"]), upsample_kernel_sizes=list(cfg["upsample_kernel_sizes"]),
                          resblock_kernel_sizes=list(cfg["resblock_kernel_sizes"]),
                          resblock_dilation_sizes=[list(d) for d in cfg["resblock_dilation_sizes"]]))
    gen = ref.Generator(h)
    gen.remove_weight_norm()                      # inference.py:60
    gen.load_state_dict(sd, strict=True)
    gen.eval()
    with torch.no_grad():
        return gen(mel)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref = ref_loader.load_hifigan()
    out = {}
    for tag, cfg, seed, B, T in (("v1", H.V1, 0, 1, 12), ("small", H.SMALL, 1, 2, 21), ("rb2", H.SMALL_RB2, 2, 2, 9)):
        sd = H.make_state(cfg, seed=seed)
        mel = H.make_mel(B, T, seed=seed + 10)
        wav = run_reference(ref, cfg, sd, mel)
        out[tag + "_mel"] = mel.numpy()
        out[tag + "_wav"] = wav.numpy()
        out[tag + "_seed"] = seed
        out[tag + "_wsum"] = float(sum(float(v.double().abs().sum()) for v in sd.values()))
        print(tag, tuple(wav.shape), "max |wav| %.3f" % float(wav.abs().max()))
    np.savez_compressed(os.path.join(OUT, "hifigan.npz"), **out)
    print("written", os.path.join(OUT, "hifigan.npz"))


if __name__ == "__main__":
    main()
