"""GPU parity tests (-m gpu) of the HiFi-GAN generator kernels (csrc/voc.hip) through the C ABI against the CPU oracle
and the reference's golden outputs.  Tolerance: split-bf16 contractions with fp32 accumulate -> max|err| <= 1e-4 * max|ref|
on every intermediate scale; the waveform itself (|wav| <= 1) additionally <= 1e-4 absolute."""
import importlib
import warnings

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import hifigan_oracle as H

pytestmark = pytest.mark.gpu
REL = 1e-4
CFGS = {"v1": H.V1, "small": H.SMALL, "rb2": H.SMALL_RB2}


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available()
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("tag", ["v1", "small", "rb2"])
def test_vocoder_matches_reference_golden(S, dev, tag):
    g = golden("hifigan.npz")
    cfg = CFGS[tag]
    sd = H.make_state(cfg, seed=int(g[tag + "_seed"]))
    voc = S.Vocoder(**cfg)
    wav = voc.forward(voc.pack(sd, dev), torch.from_numpy(g[tag + "_mel"]).to(dev)).cpu()
    ref = torch.from_numpy(g[tag + "_wav"])
    assert wav.shape == ref.shape
    print("%s: max|ref| %.3f  max|err| %.2e" % (tag, float(ref.abs().max()), float((wav - ref).abs().max())))
    assert relerr(wav, ref) <= REL and float((wav - ref).abs().max()) <= 1e-4


@pytest.mark.parametrize("B,T", [(1, 1), (3, 37), (2, 130)])
def test_vocoder_matches_oracle_odd_shapes(S, dev, B, T):
    """Lengths that are not multiples of any tile (edge padding of every dilated kernel, partial tiles), strong signal
    (gain 3: the output tanh saturates for part of the samples)."""
    cfg = H.SMALL
    sd = H.make_state(cfg, seed=5, gain=1.0)
    sd["conv_post.weight"] = sd["conv_post.weight"] * 6.0
    mel = H.make_mel(B, T, seed=T)
    ref = H.generator_forward(sd, cfg, mel)
    voc = S.Vocoder(**cfg)
    wav = voc.forward(voc.pack(sd, dev), mel.to(dev)).cpu()
    assert float(ref.abs().max()) > 0.3
    assert float((wav - ref).abs().max()) <= 1e-4


def test_vocoder_v1_full_length_properties(S, dev):
    """The shape inference.py feeds it at the bench size (80 x 1024 frames -> 262 144 samples), V1 width: finite, in
    (-1, 1), run-to-run bit-identical, batch entries independent; one utterance against the CPU oracle (~10 s)."""
    sd = H.make_state(H.V1, seed=0)
    voc = S.Vocoder(**H.V1)
    blob = voc.pack(sd, dev)
    mel = H.make_mel(2, 1024, seed=9)
    a = voc.forward(blob, mel.to(dev))
    b = voc.forward(blob, mel.to(dev))
    assert a.shape == (2, 1, 262144) and torch.isfinite(a).all() and float(a.abs().max()) < 1.0
    assert torch.equal(a, b)
    one = voc.forward(blob, mel[1:2].contiguous().to(dev))
    assert torch.equal(one, a[1:2])
    ref = H.generator_forward(sd, H.V1, mel[1:2])
    assert float((one.cpu() - ref).abs().max()) <= 1e-4


def test_generator_module_drop_in(S, dev):
    """`Generator(h)` exactly as inference.py:57-61 uses it: load a weight-normalised checkpoint, remove_weight_norm(),
    .cuda(), forward under no_grad -> HIP kernels; also with weight norm still attached (folded on the fly)."""
    warnings.simplefilter("ignore")
    M = importlib.import_module("speech-backbones_amd.hifi_gan.models")
    env = importlib.import_module("speech-backbones_amd.hifi_gan.env")
    cfg = H.SMALL
    h = env.AttrDict(dict(cfg, upsample_rates=list(cfg["upsample_rates"]), upsample_kernel_sizes=list(cfg["upsample_kernel_sizes"]),
                          resblock_kernel_sizes=list(cfg["resblock_kernel_sizes"]),
                          resblock_dilation_sizes=[list(d) for d in cfg["resblock_dilation_sizes"]]))
    torch.manual_seed(1)
    gen = M.Generator(h)
    with torch.no_grad():
        for p in gen.parameters():                      # init std 0.01 is tiny: scale the direction tensors up
            if p.dim() == 3:
                p.mul_(30.0)
    mel = H.make_mel(2, 19, seed=2)
    want = gen._forward_torch(mel).detach()
    gen = gen.to(dev).eval()
    with torch.no_grad():
        got_wn = gen(mel.to(dev)).cpu()                 # weight_g / weight_v still attached
    assert float((got_wn - want).abs().max()) <= 1e-4
    gen.remove_weight_norm()
    with torch.no_grad():
        got = gen(mel.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 1e-4
