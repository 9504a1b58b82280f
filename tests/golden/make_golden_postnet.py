"""Golden vector of DiffVC's PostNet, produced by the REFERENCE's own module (DiffVC/model/postnet.py) where /root/reference is
mounted:  python tests/golden/make_golden_postnet.py.  Weights: oracle.postnet_oracle.make_state(seed) (not stored)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import postnet_oracle as P  # noqa: E402
from oracle import ref_loader  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(4)
    ref = ref_loader.load_diffvc()
    sd = P.make_state(128, seed=0)
    net = ref.postnet.PostNet(128).eval()
    net.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 80, 45, generator=g)
    lens = torch.tensor([45, 28])
    mask = (torch.arange(45).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(1).float()
    with torch.no_grad():
        y = net(x, mask)
    np.savez_compressed(os.path.join(OUT, "postnet.npz"), x=x.numpy(), mask=mask.numpy(), y=y.numpy(),
                        wsum=float(sum(float(v.double().abs().sum()) for v in sd.values())))
    print("written", os.path.join(OUT, "postnet.npz"), tuple(y.shape))


if __name__ == "__main__":
    main()
