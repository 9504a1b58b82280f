"""CPU: the encoders of SURVEY 8f rank 4 -- oracle vs the reference's golden outputs and (live) the reference modules; the
product modules keep the reference's state_dict; the C-ABI plan expects exactly that layout."""
import importlib

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import encoder_oracle as E
from oracle import ref_loader


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_text_encoder_oracle_matches_reference_golden():
    g = golden("encoder.npz")
    sd = E.make_state("text", seed=0)
    assert abs(float(sum(float(v.double().abs().sum()) for v in sd.values())) - float(g["text_wsum"])) <= 1e-6 * float(g["text_wsum"])
    mu, logw, mask = E.text_encoder_forward(sd, _t(g["text_ids"]), _t(g["text_lens"]))
    assert torch.allclose(mu, _t(g["text_mu"]), atol=2e-5) and torch.allclose(logw, _t(g["text_logw"]), atol=2e-5)


def test_mel_encoder_oracle_matches_reference_golden():
    g = golden("encoder.npz")
    sd = E.make_state("mel", seed=2)
    out = E.mel_encoder_forward(sd, _t(g["mel_in"]), _t(g["mel_mask"]))
    assert torch.allclose(out, _t(g["mel_out"]), atol=2e-5)


def test_product_modules_share_the_layout_and_match_the_oracle():
    S = importlib.import_module("speech-backbones_amd")
    TE = importlib.import_module("speech-backbones_amd.model.text_encoder")
    ME = importlib.import_module("speech-backbones_amd.diffvc.model.encoder")
    sd = E.make_state("text", seed=3)
    enc = TE.TextEncoder(149, 80, 192, 768, 256, 2, 6, 3, 0.1, 4).eval()
    enc.load_state_dict(sd, strict=True)
    assert [k for k, _ in S.Encoder("text").param_layout()] == list(enc.state_dict().keys())
    ids = torch.randint(0, 149, (2, 23))
    lens = torch.tensor([23, 9])
    with torch.no_grad():
        mu, logw, mask = enc(ids, lens)                       # CPU tensors: torch composition
    mu_o, logw_o, _ = E.text_encoder_forward(sd, ids, lens)
    assert torch.allclose(mu, mu_o, atol=2e-5) and torch.allclose(logw, logw_o, atol=2e-5)
    sdm = E.make_state("mel", seed=4)
    menc = ME.MelEncoder(80, 192, 768, 2, 6, 3, 0.1, window_size=4).eval()
    menc.load_state_dict(sdm, strict=True)
    assert sorted(k for k, _ in S.Encoder("mel", 0, 80, 192, 768, 0, 2, 6, 3, 4).param_layout()) == sorted(menc.state_dict().keys())
    mel = torch.randn(2, 80, 30)
    mm = E.sequence_mask(torch.tensor([30, 11]), 30).unsqueeze(1).float()
    with torch.no_grad():
        assert torch.allclose(menc(mel, mm), E.mel_encoder_forward(sdm, mel, mm), atol=2e-5)


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
def test_oracle_against_reference_live():
    ref = ref_loader.load_gradtts()
    sd = E.make_state("text", seed=5)
    enc = ref.text_encoder.TextEncoder(149, 80, 192, 768, 256, 2, 6, 3, 0.1, 4).eval()
    enc.load_state_dict(sd, strict=True)
    ids = torch.randint(0, 149, (2, 50))
    lens = torch.tensor([50, 1])
    with torch.no_grad():
        mu, logw, mask = enc(ids, lens)
    mu_o, logw_o, mask_o = E.text_encoder_forward(sd, ids, lens)
    assert torch.equal(mask_o, mask) and torch.allclose(mu, mu_o, atol=1e-6) and torch.allclose(logw, logw_o, atol=1e-6)


def test_postnet_oracle_module_and_layout():
    """DiffVC PostNet (postnet.py:40-53): oracle vs the reference's golden output, product module state_dict == plan layout,
    torch composition == oracle."""
    from oracle import postnet_oracle as P
    g = golden("postnet.npz")
    sd = P.make_state(128, seed=0)
    assert abs(float(sum(float(v.double().abs().sum()) for v in sd.values())) - float(g["wsum"])) <= 1e-6 * float(g["wsum"])
    y = P.postnet_forward(sd, _t(g["x"]), _t(g["mask"]))
    assert torch.allclose(y, _t(g["y"]), atol=2e-5)
    S = importlib.import_module("speech-backbones_amd")
    PN = importlib.import_module("speech-backbones_amd.diffvc.model.postnet")
    net = PN.PostNet(128).eval()
    net.load_state_dict(sd, strict=True)
    assert [k for k, _ in S.PostNetPlan(128).param_layout()] == list(net.state_dict().keys())
    with torch.enable_grad():
        assert torch.allclose(net(_t(g["x"]), _t(g["mask"])), _t(g["y"]), atol=2e-5)
