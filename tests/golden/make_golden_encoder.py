"""Golden vectors of the encoders, produced by the REFERENCE's own modules (Grad-TTS/model/text_encoder.py TextEncoder,
DiffVC/model/encoder.py MelEncoder) where /root/reference is mounted:  python tests/golden/make_golden_encoder.py
Weights come from oracle.encoder_oracle.make_state(seed) (deterministic, not stored)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoder_oracle as E  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(4)
    ref = ref_loader.load_gradtts()
    sd = E.make_state("text", seed=0)
    enc = ref.text_encoder.TextEncoder(149, 80, 192, 768, 256, 2, 6, 3, 0.1, 4).eval()
    enc.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 149, (3, 29), generator=g)
    lens = torch.tensor([29, 17, 3])
    with torch.no_grad():
        mu, logw, mask = enc(ids, lens)
    out = {"text_ids": ids.numpy(), "text_lens": lens.numpy(), "text_mu": mu.numpy(), "text_logw": logw.numpy(),
           "text_wsum": float(sum(float(v.double().abs().sum()) for v in sd.values()))}
    vc = ref_loader.load_diffvc()
    encmod = vc.encoder
    sdm = E.make_state("mel", seed=2)
    menc = encmod.MelEncoder(80, 192, 768, 2, 6, 3, 0.1, window_size=4).eval()
    menc.load_state_dict(sdm, strict=True)
    mel = torch.randn(2, 80, 41, generator=g)
    mmask = E.sequence_mask(torch.tensor([41, 26]), 41).unsqueeze(1).float()
    with torch.no_grad():
        mout = menc(mel, mmask)
    out.update({"mel_in": mel.numpy(), "mel_mask": mmask.numpy(), "mel_out": mout.numpy(),
                "mel_wsum": float(sum(float(v.double().abs().sum()) for v in sdm.values()))})
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)
    print("written", os.path.join(OUT, "encoder.npz"), mu.shape, mout.shape)


if __name__ == "__main__":
    main()
