"""Range contract of GTTS_PREC_F16F8 (-m gpu; include/gradtts_abi.h, ABI 6; round-5 review item 3).

The reference arithmetic is plain fp32 (Grad-TTS/model/diffusion.py:49-58): any weight, any activation.  The f16 + fp8 split has
two limits, and neither may be silent:
  (a) a 3x3 Block-convolution (or Upsample) weight with |w| >= 63.97 does not fit fp16(w 2^10): gtts_pack_weights checks on the device and returns
      GTTS_E_RANGE (nothing is packed as inf); the drop-in module then samples in bf16x3 and warns;
  (b) an activation with |x| >= 1024 keeps only an fp16-grade cross term: the staging kernels count such events and the maximum |x|
      into the first 16 bytes of the workspace, read with gtts_workspace_status / Plan.range_status / GradLogPEstimator2d.range_status.
"""
import importlib
import warnings

import pytest
import torch

from oracle import gradtts_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _big_weight_state(value=70.0, layer="downs.1.0.block2.block.0.weight"):
    sd = dict(O.make_estimator_state(seed=0))
    w = sd[layer].clone()
    w[3, 5, w.shape[2] // 2, w.shape[3] - 1] = value
    sd[layer] = w
    return sd


@pytest.mark.parametrize("value", [70.0, -1e4, float("inf")])
def test_pack_refuses_out_of_range_weights(S, dev, value):
    sd = _big_weight_state(value)
    plan = S.Plan(precision=S.PREC_F16F8)
    with pytest.raises(S.RangeError) as ei:
        plan.pack(sd, dev)
    msg = str(ei.value)
    assert "(-7)" in msg and "1 weight(s) out of range" in msg and "block2.block.0.weight" in msg and "BF16X3" in msg, msg
    # the same parameters pack in bf16x3, which has no such limit, and in f16f8 once the weight is inside the range
    S.Plan(precision=S.PREC_BF16X3).pack(sd, dev)
    S.Plan(precision=S.PREC_F16F8).pack(_big_weight_state(63.9), dev)
    # a layer the split does not touch (1x1 res_conv: bf16x3 in every precision) may hold anything
    S.Plan(precision=S.PREC_F16F8).pack(_big_weight_state(500.0, "downs.1.0.res_conv.weight"), dev)


def test_pack_refuses_out_of_range_upsample_weight(S, dev):
    """The Upsample layers take the split as well (conv_up.hip): same check, the message names the layer."""
    sd = _big_weight_state(-80.0, "ups.0.3.conv.weight")
    with pytest.raises(S.RangeError) as ei:
        S.Plan(precision=S.PREC_F16F8).pack(sd, dev)
    assert "ups.0.3.conv.weight" in str(ei.value), str(ei.value)
    S.Plan(precision=S.PREC_F16F8).pack(_big_weight_state(-63.9, "ups.0.3.conv.weight"), dev)
    # Downsample stays bf16x3
    S.Plan(precision=S.PREC_F16F8).pack(_big_weight_state(500.0, "downs.0.3.conv.weight"), dev)


def test_weight_at_the_edge_is_exact_enough(S, dev):
    """|w| = 63.9 is inside the format: the estimator still matches the oracle at the usual bound."""
    sd = _big_weight_state(63.9)
    inp = O.make_inputs(2, 64, seed=3)
    t = torch.tensor([0.3, 0.7])
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t)
    plan = S.Plan(precision=S.PREC_F16F8)
    out = plan.estimator_forward(plan.pack(sd, dev), inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev)).cpu()
    assert torch.isfinite(out).all()
    assert relerr(out, ref) <= 1e-4


def test_module_falls_back_to_bf16x3_and_warns(S, dev):
    M = importlib.import_module("speech-backbones_amd.model.diffusion")
    sd = _big_weight_state(70.0)
    dec = M.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.load_state_dict({"estimator." + k: v for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    inp = O.make_inputs(2, 40, seed=6)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = dec(inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 3).cpu()
    assert any(issubclass(w.category, RuntimeWarning) and "bf16x3" in str(w.message) for w in rec), [str(w.message) for w in rec]
    assert dec.estimator._precision == S.PREC_BF16X3
    assert relerr(out, O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 3)) <= 1e-4
    # an explicit request is not overridden: the error is the caller's
    dec2 = M.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec2.load_state_dict({"estimator." + k: v for k, v in sd.items()}, strict=True)
    dec2 = dec2.to(dev).eval()
    dec2.estimator.set_precision("f16f8")
    with pytest.raises(S.RangeError):
        dec2(inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 3)


@pytest.mark.parametrize("conv_ws", [False, True], ids=["conv_mfma", "conv_ws"])
def test_activation_range_record(S, dev, conv_ws):
    sd = O.make_estimator_state(seed=0)
    plan = S.Plan(precision=S.PREC_F16F8, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    inp = O.make_inputs(2, 64, seed=9)
    t = torch.tensor([0.4, 0.6])
    z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
    plan.estimator_forward(blob, z, m, mu, t.to(dev))
    assert plan.range_status() == (0, 0.0)
    # scaled inputs: the residual stream entering the mask-prologue convolutions of the split layers passes |x| = 1024 (but not the
    # fp16 half's 65504); which scale does it depends on which layers take the split (conv_ws: also the 64-channel ones)
    for scale in (50.0, 60.0, 70.0, 80.0, 90.0, 100.0, 150.0, 200.0):
        out = plan.estimator_forward(blob, z * scale, m, mu * scale, t.to(dev))
        ev, mx = plan.range_status()
        print("activation range record at %gx inputs: %d events, max |x| = %.1f" % (scale, ev, mx))
        if ev > 0:
            break
    assert ev > 0 and mx >= 1024.0
    if mx < 65504.0:                 # (beyond it the fp16 half itself overflows: documented, and exactly what the record is for)
        assert torch.isfinite(out).all()
    # sticky over the step ranges of one sampling run, reset by the next run
    plan.reverse_diffusion(blob, z * scale, m, mu * scale, 2)
    ev2, mx2 = plan.range_status()
    assert ev2 > 0 and mx2 >= 1024.0
    plan.reverse_diffusion(blob, z, m, mu, 2)
    assert plan.range_status() == (0, 0.0)
    # the other precisions never record
    p3 = S.Plan(precision=S.PREC_BF16X3)
    p3.estimator_forward(p3.pack(sd, dev), z * scale, m, mu * scale, t.to(dev))
    assert p3.range_status() == (0, 0.0)


def test_module_range_status(S, dev):
    M = importlib.import_module("speech-backbones_amd.model.diffusion")
    sd = O.make_estimator_state(seed=2)
    dec = M.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    dec.load_state_dict({"estimator." + k: v for k, v in sd.items()}, strict=True)
    dec = dec.to(dev).eval()
    inp = O.make_inputs(1, 64, seed=4)
    dec(inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 2)
    assert dec.estimator.range_status() == (0, 0.0)
    for scale in (50.0, 100.0, 200.0, 400.0, 800.0, 1600.0):
        dec(inp["z"].to(dev) * scale, inp["mask"].to(dev), inp["mu"].to(dev) * scale, 2)
        ev, mx = dec.estimator.range_status()
        if ev > 0:
            break
    assert ev > 0 and mx >= 1024.0


@pytest.mark.parametrize("prec", ["f16f8", "bf16x3", "f16f8_mid"])
def test_results_do_not_depend_on_what_the_workspace_held(S, dev, prec):
    """Found in round 6: the persistent convolution's 16-byte halo loads read a few bytes beside a tensor (masked out by a factor 0), and
    0 x NaN = NaN -- after ONE call that overflowed (NaN in the workspace) every later call on that workspace returned NaN, and a
    workspace that happened to be allocated over NaN bit patterns would have done the same.  The mask factor is now applied with
    v_mul_legacy_f32 (0 x anything = 0): a workspace filled with NaN, then with Inf, must give the bit-identical result."""
    sd = O.make_estimator_state(seed=2)
    kw = dict(precision=S.PREC_F16F8) if prec == "f16f8" else (dict(precision=S.PREC_BF16X3) if prec == "bf16x3" else dict(precision=S.PREC_F16F8, conv_ws=False, streams=3))
    plan = S.Plan(**kw)
    blob = plan.pack(sd, dev)
    inp = O.make_inputs(3, 100, seed=4, ragged=True)
    z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
    t = torch.tensor([0.75, 0.3, 0.5]).to(dev)
    want = plan.estimator_forward(blob, z, m, mu, t).clone()
    want_rd = plan.reverse_diffusion(blob, z, m, mu, 2).clone()
    assert torch.isfinite(want).all() and torch.isfinite(want_rd).all()
    ws = list(plan._ws.values())[0]
    for poison in (float("nan"), float("inf"), -1e38):
        ws.view(torch.float32).fill_(poison)
        assert torch.equal(plan.estimator_forward(blob, z, m, mu, t), want), poison
        ws.view(torch.float32).fill_(poison)
        assert torch.equal(plan.reverse_diffusion(blob, z, m, mu, 2), want_rd), poison
    # and the sequence that exposed it: an overflowing call, then a normal one
    plan.estimator_forward(blob, z * 1000.0, m, mu * 1000.0, t)
    assert torch.equal(plan.estimator_forward(blob, z, m, mu, t), want)
