"""CPU: HiFi-GAN generator -- oracle vs the reference's golden outputs and (live) the reference's own
module; the product's drop-in `hifi_gan` package keeps the reference's state_dict with and without weight norm."""
import importlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from oracle import hifigan_oracle as H
from oracle import ref_loader

CFGS = {"v1": H.V1, "small": H.SMALL, "rb2": H.SMALL_RB2}


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["v1", "small", "rb2"])
def test_oracle_matches_reference_golden(tag):
    g = golden("hifigan.npz")
    sd = H.make_state(CFGS[tag], seed=int(g[tag + "_seed"]))
    wsum = float(sum(float(v.double().abs().sum()) for v in sd.values()))
    assert abs(wsum - float(g[tag + "_wsum"])) <= 1e-6 * wsum
    wav = H.generator_forward(sd, CFGS[tag], _t(g[tag + "_mel"]))
    assert wav.shape == g[tag + "_wav"].shape
    assert torch.allclose(wav, _t(g[tag + "_wav"]), rtol=0, atol=2e-6)


def _attr(cfg):
    env = importlib.import_module("speech-backbones_amd.hifi_gan.env")
    return env.AttrDict(dict(cfg, upsample_rates=list(cfg["upsample_rates"]), upsample_kernel_sizes=list(cfg["upsample_kernel_sizes"]),
                             resblock_kernel_sizes=list(cfg["resblock_kernel_sizes"]),
                             resblock_dilation_sizes=[list(d) for d in cfg["resblock_dilation_sizes"]]))


def test_drop_in_generator_state_dict_and_torch_path():
    """Same parameter names / shapes as the golden layout after remove_weight_norm(); the autograd composition (training
    path) equals the oracle; inference on a CPU tensor raises instead of falling back."""
    warnings.simplefilter("ignore")
    M = importlib.import_module("speech-backbones_amd.hifi_gan.models")
    for tag in ("small", "rb2"):
        cfg = CFGS[tag]
        gen = M.Generator(_attr(cfg))
        keys_wn = list(gen.state_dict().keys())
        assert "conv_pre.weight_g" in keys_wn and "conv_pre.weight_v" in keys_wn
        gen.remove_weight_norm()
        sd = H.make_state(cfg, seed=3)
        assert sorted(gen.state_dict().keys()) == sorted(sd.keys())
        gen.load_state_dict(sd, strict=True)
        mel = H.make_mel(1, 7, seed=4)
        out = gen(mel)                                   # parameters require grad -> torch composition
        assert torch.allclose(out, H.generator_forward(sd, cfg, mel), atol=1e-6)
        with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
            gen(mel)
    # the host-side plan expects exactly the folded parameters, in module order
    S = importlib.import_module("speech-backbones_amd")
    v = S.Vocoder(**H.V1)
    layout = v.param_layout()
    sd = H.make_state(H.V1, seed=0)
    assert sorted(k for k, _ in layout) == sorted(sd.keys())
    assert all(tuple(sd[k].shape) == s for k, s in layout)
    assert v.hop == 256
    with open(os.path.join(ROOT, "tests", "golden", "hifigan-config.json")) as f:
        h = json.load(f)
    assert S.Vocoder.from_config(h).param_layout() == layout


@pytest.mark.skipif(not os.path.isdir("/root/reference/Grad-TTS/hifi-gan"), reason="/root/reference not mounted")
def test_oracle_and_module_against_reference_live():
    warnings.simplefilter("ignore")
    ref = ref_loader.load_hifigan()
    M = importlib.import_module("speech-backbones_amd.hifi_gan.models")
    cfg = H.SMALL
    torch.manual_seed(0)
    r = ref.Generator(ref.AttrDict(_attr(cfg)))
    m = M.Generator(_attr(cfg))
    assert [(k, tuple(v.shape)) for k, v in r.state_dict().items()] == [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    m.load_state_dict(r.state_dict(), strict=True)       # checkpoint format: weight_g / weight_v
    mel = H.make_mel(2, 9, seed=1)
    with torch.no_grad():
        want = r(mel)
    got = m(mel)                                         # torch composition with weight norm attached
    assert torch.allclose(got, want, atol=1e-6)
    eff = m._effective_state()                           # what the HIP packer receives
    r.remove_weight_norm()
    for k, v in r.state_dict().items():
        assert torch.allclose(eff[k], v, atol=1e-7), k
    assert torch.allclose(H.generator_forward(dict(r.state_dict()), cfg, mel), want, atol=1e-6)
