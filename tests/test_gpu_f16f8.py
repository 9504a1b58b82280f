"""GPU parity of GTTS_PREC_F16F8 (-m gpu): the 3x3 Block convolutions as fp16 hi*hi + ONE fp8 MFMA for both cross terms
(csrc/common.h, csrc/conv_mfma.hip NSPLIT == 3; Grad-TTS/model/diffusion.py:49-58 is the op).

The bounds are the ones the bf16x3 mode is held to (tests/test_gpu_parity*.py): 1e-4 of max|ref| per estimator call and per
tap, 1e-4 relative / 1e-3 max-abs on the mel-scale fixture for the free-running N = 50 sampler at T = 1024 -- the north
star's tolerance.  Expected (tests/numerics_emul.py, tools/probe/f8_probe2.hip): ~3x the bf16x3 error, i.e. ~5e-5 per call
and ~5e-4 at N = 50.  Also: results do not depend on how utterances are batched (bit-identical), masked frames are exactly
zero, activations beyond the fp8 operand's range degrade gracefully (finite, fp16-grade), not to NaN.
"""
import importlib

import pytest
import torch

from conftest import oracle_n50
from oracle import gradtts_oracle as O

pytestmark = pytest.mark.gpu
REL = 1e-4


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


RAW = ["downs.0.0", "downs.0.1", "downs.1.0", "downs.1.1", "downs.2.0", "downs.2.1", "mid_block1", "mid_block2",
       "ups.0.0", "ups.0.1", "ups.1.0", "ups.1.1"]


both_convs = pytest.mark.parametrize("conv_ws", [False, True], ids=["conv_mfma", "conv_ws"])


@both_convs
@pytest.mark.parametrize("n_spks,B,T", [(1, 2, 64), (4, 3, 100)])
def test_every_op_output_matches_oracle_taps(S, dev, n_spks, B, T, conv_ws):
    """Every Block conv output (*.raw: both prologues, 64- and 128-channel tiles, the concatenated up-path inputs), tails,
    attention, resampling against the oracle's taps of the same call."""
    sd = O.make_estimator_state(seed=0, n_spks=n_spks)
    inp = O.make_inputs(B, T, seed=1234, spk_dim=64 if n_spks > 1 else None)
    t = torch.linspace(0.15, 0.9, B)
    taps = {}
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t, inp.get("spk"), taps=taps)
    plan = S.Plan(n_spks=n_spks, keep_intermediates=True, precision=S.PREC_F16F8, conv_ws=conv_ws)
    assert plan.conv_ws is conv_ws
    blob = plan.pack(sd, dev)
    out = plan.estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev),
                                 inp["spk"].to(dev) if n_spks > 1 else None).cpu()
    hip = {k: v.detach().cpu().clone() for k, v in plan.tensors(B, T, dev).items()}
    worst, checked = ("", 0.0), 0
    for name, want in taps.items():
        if name == "t_emb" or name.endswith(".tb") or name == "est" or name not in hip:
            continue
        e = relerr(hip[name], want)
        checked += 1
        if e > worst[1]:
            worst = (name, e)
    assert checked == 12 * 3 + 6 + 2 + 2 + 1 + 1, checked
    print("f16f8 worst tap: %s rel %.2e; estimator output rel %.2e" % (worst[0], worst[1], relerr(out, ref)))
    assert worst[1] <= REL, worst
    assert relerr(out, ref) <= REL


@both_convs
def test_local_block_conv_on_hip_inputs(S, dev, conv_ws):
    """The 3x3 convolution alone: oracle conv2d applied to the HIP path's OWN input of the layer (so the error of the split is
    not mixed with its producers'), for a mask-prologue and a GroupNorm-prologue layer of every tile shape."""
    import torch.nn.functional as F
    sd = O.make_estimator_state(seed=3)
    B, T = 2, 96
    inp = O.make_inputs(B, T, seed=5, ragged=True)
    t = torch.tensor([0.3, 0.8])
    plan = S.Plan(keep_intermediates=True, precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    plan.estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev))
    torch.cuda.synchronize()
    hip = {k: v.detach().cpu().clone() for k, v in plan.tensors(B, T, dev).items()}
    t_emb = O.time_mlp(sd, t, 64, 1000)
    masks = {0: inp["mask"].unsqueeze(1)}
    masks[1] = masks[0][:, :, :, ::2]
    masks[2] = masks[1][:, :, :, ::2]
    worst = 0.0
    for name, lvl in (("downs.0.1", 0), ("downs.1.1", 1), ("downs.2.1", 2), ("mid_block2", 2), ("ups.1.1", 1)):
        m = masks[lvl]
        # block2: input = Mish(GN(b1.raw)) * mask + time bias, the GroupNorm prologue of the kernel
        raw1 = hip[name + ".b1.raw"]
        y = F.group_norm(raw1, 8, sd[name + ".block1.block.1.weight"], sd[name + ".block1.block.1.bias"], eps=1e-5)
        h = O.mish(y) * m + F.linear(O.mish(t_emb), sd[name + ".mlp.1.weight"], sd[name + ".mlp.1.bias"]).unsqueeze(-1).unsqueeze(-1)
        want = F.conv2d(h * m, sd[name + ".block2.block.0.weight"], sd[name + ".block2.block.0.bias"], padding=1)
        e = relerr(hip[name + ".b2.raw"], want)
        worst = max(worst, e)
        print("%s.block2 (GroupNorm prologue) rel %.2e" % (name, e))
    print("f16f8 local conv worst rel %.2e" % worst)
    assert worst <= 6e-5


@both_convs
def test_reverse_diffusion_n50_t1024_vs_oracle(S, dev, conv_ws):
    """The headline configuration's own N and T, one full and one ragged utterance (the scale-350 fixture: block inputs reach
    |x| = 450, tests/numerics_emul.py)."""
    sd, inp, ref = oracle_n50("scale350")
    plan = S.Plan(precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 50).cpu()
    assert torch.isfinite(out).all()
    assert float((out * (1 - inp["mask"])).abs().max()) == 0.0
    err = relerr(out, ref)
    print("f16f8 N=50 T=1024: max|ref| %.4g  max|err| %.3e  rel %.2e" % (float(ref.abs().max()), float((out - ref).abs().max()), err))
    assert err <= REL


@both_convs
def test_reverse_diffusion_n50_t1024_mel_scale_abs(S, dev, conv_ws):
    """Same N and T on the mel-scale fixture: the north star's literal 1e-3 max-abs."""
    sd, inp, ref = oracle_n50("melscale")
    plan = S.Plan(precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 50).cpu()
    err = float((out - ref).abs().max())
    print("f16f8 mel-scale N=50: max|ref| %.4g  max|err| %.3e  margin to 1e-3: %.1fx" % (float(ref.abs().max()), err, 1e-3 / max(err, 1e-30)))
    assert 1.0 < float(ref.abs().max()) < 20
    assert err <= 1e-3


@pytest.mark.parametrize("prec", ["f16f8", "bf16x3"])
def test_reverse_diffusion_n50_t1024_mel_scale_attention_on(S, dev, prec):
    """The harder fixture (round-5 review 3c): Rezero.g = 0.15, so the LinearAttention branch (Grad-TTS/model/diffusion.py:82-110)
    is ~15 % of every residual it joins instead of 2 %; N = 50, T = 1024, the north star's literal 1e-3 max-abs, in the drop-in
    modules' default precision and in bf16x3 (same bound; the margin of each is printed)."""
    sd, inp, ref = oracle_n50("melscale_attn")
    plan = S.Plan(precision=S.PREC_F16F8 if prec == "f16f8" else S.PREC_BF16X3)
    blob = plan.pack(sd, dev)
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 50).cpu()
    err = float((out - ref).abs().max())
    print("%s mel-scale N=50, Rezero.g = 0.15: max|ref| %.4g  max|err| %.3e  margin to 1e-3: %.1fx" % (prec, float(ref.abs().max()), err, 1e-3 / max(err, 1e-30)))
    assert torch.isfinite(out).all()
    assert 1.0 < float(ref.abs().max()) < 40
    assert err <= 1e-3


@both_convs
def test_batching_does_not_change_results(S, dev, conv_ws):
    """B = 16 == 8 + 8 and an utterance alone (half-height tiles) == the same utterance inside a batch, bit for bit; run to run
    reproducible; masked frames exactly zero."""
    sd = O.make_estimator_state(seed=0)
    plan = S.Plan(precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    B, T = 16, 1024
    inp = O.make_inputs(B, T, seed=1234, ragged=True)
    z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
    a = plan.reverse_diffusion(blob, z, m, mu, 2)
    b = plan.reverse_diffusion(blob, z, m, mu, 2)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)
    assert float((a * (1 - m)).abs().max()) == 0.0
    lo = plan.reverse_diffusion(blob, z[:8].contiguous(), m[:8].contiguous(), mu[:8].contiguous(), 2)
    hi = plan.reverse_diffusion(blob, z[8:].contiguous(), m[8:].contiguous(), mu[8:].contiguous(), 2)
    assert torch.equal(a, torch.cat([lo, hi], 0))
    for i in (0, 5, 15):
        one = plan.reverse_diffusion(blob, z[i:i + 1].contiguous(), m[i:i + 1].contiguous(), mu[i:i + 1].contiguous(), 2)
        assert torch.equal(one, a[i:i + 1]), "utterance %d differs between B=1 and B=%d" % (i, B)
    ref = O.reverse_diffusion(sd, inp["z"][3:4], inp["mask"][3:4], inp["mu"][3:4], 2)
    assert relerr(a[3:4].cpu(), ref) <= REL


@both_convs
def test_large_activations_degrade_gracefully(S, dev, conv_ws):
    """Inputs 50x the fixture's scale drive Block inputs past the fp8 operands' range (|x| > 1024: q8(xl 2^S) saturates, |x| > 7168:
    q8(x 2^-D) as well) but not past the fp16 half's (65504): the result stays finite and fp16-grade (the hi*hi term is unaffected),
    never NaN -- v_cvt_pk_fp8_f32 returns NaN beyond 448, so both operands go through v_med3_f32 first (common.h)."""
    sd = O.make_estimator_state(seed=0)
    plan = S.Plan(precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    inp = O.make_inputs(2, 64, seed=9)
    z, mu = inp["z"] * 50.0, inp["mu"] * 50.0
    t = torch.tensor([0.4, 0.6])
    ref = O.estimator_forward(sd, z, inp["mask"], mu, t)
    out = plan.estimator_forward(blob, z.to(dev), inp["mask"].to(dev), mu.to(dev), t.to(dev)).cpu()
    assert torch.isfinite(out).all()
    e = relerr(out, ref)
    print("f16f8 with 50x inputs (max|z| %.0f): rel %.2e" % (float(z.abs().max()), e))
    assert e <= 2e-3


def test_diffusion_module_set_precision(S, dev):
    M = importlib.import_module("speech-backbones_amd.model.diffusion")
    sd = O.make_estimator_state(seed=11)
    dec = M.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.load_state_dict({"estimator." + k: v for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    dec.estimator.set_precision("f16f8")
    inp = O.make_inputs(2, 40, seed=6)
    out = dec(inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 5).cpu()
    assert relerr(out, O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 5)) <= REL


def test_module_batch_buckets_agree_to_fp32_grade_and_pin_is_exact(S, dev):
    """The drop-in module picks its plan by batch size in f16f8 (advisor, round 5): B = 1 / B >= 7 persistent kernel, B = 2...6
    uniform waves with the 64-channel layers in bf16x3.  Across buckets the same utterance agrees to fp32-grade rounding; with the
    variant pinned it is bit-identical at every batch size."""
    M = importlib.import_module("speech-backbones_amd.model.diffusion")
    sd = O.make_estimator_state(seed=5)
    dec = M.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.load_state_dict({"estimator." + k: v for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    inp = O.make_inputs(8, 128, seed=8, ragged=True)
    z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
    full = dec(z, m, mu, 10)                                     # B = 8: 'main'
    four = dec(z[:4].contiguous(), m[:4].contiguous(), mu[:4].contiguous(), 10)      # B = 4: 'mid'
    one = dec(z[:1].contiguous(), m[:1].contiguous(), mu[:1].contiguous(), 10)       # B = 1: 'main'
    assert torch.equal(one, full[:1])                            # same bucket: bit-identical
    e = relerr(four, full[:4])
    print("f16f8 module, B = 4 bucket vs B = 8 bucket after 10 steps: rel %.2e" % e)
    assert 0.0 < e <= 1e-4
    dec.estimator.pin_variant("main")
    assert torch.equal(dec(z[:4].contiguous(), m[:4].contiguous(), mu[:4].contiguous(), 10), full[:4])
    dec.estimator.pin_variant(None)
