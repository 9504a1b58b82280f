"""GPU parity tests (-m gpu) at the configurations bench.py quotes, plus kernel-level (per-op) parity.

  * config 2 at its own N: reverse diffusion N=50, T=1024 (one full, one ragged utterance) against the CPU oracle
    (SURVEY section 4 budget: max|err| <= 1e-4 * max|ref|; on a mel-scale fixture <= 1e-3 max-abs).
  * every op output of one estimator call (Block raw conv outputs, ResnetBlock tails, attention, Down/Upsample,
    GroupNorm scale/shift) against the oracle's taps, end-to-end and locally (oracle op on the HIP path's own input).
  * LinearAttention with Rezero.g = 1.0 (every other fixture attenuates the branch 50x with g = 0.02).
  * config 3: 247 speakers through GradTTS.forward(spk=...) / Diffusion(..., spk=...), plain-bf16 precision,
    N=100 teacher-forced.
  * config 4: DiffVC dim 256 at T=1024, one estimator call.
"""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import oracle_n50

from oracle import diffvc_oracle as V
from oracle import gradtts_oracle as O

pytestmark = pytest.mark.gpu
REL = 1e-4          # bf16x3 contractions, fp32 accumulate (measured ~2e-5 per call)
REL_BF16 = 2.5e-2   # plain-bf16 contractions (config 3), single estimator call (measured ~8e-3)
REL_BF16_STORE = 4e-2   # bf16 contractions AND bf16 activation storage (config 3 as written)


@pytest.fixture(scope="module")
def S():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return importlib.import_module("speech-backbones_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------------------------------------ config 2, N = 50
def test_reverse_diffusion_n50_t1024_vs_oracle(S, dev):
    """The headline configuration's own N and T: free-running 50 Euler steps on 80x1024, one full and one ragged
    utterance (Grad-TTS/model/diffusion.py:254-275).  CPU oracle: ~20 s per utterance."""
    sd, inp, ref = oracle_n50("scale350")                         # (shared with tests/test_gpu_f16f8.py: computed once per process)
    plan = S.Plan()
    blob = plan.pack(sd, dev)
    assert inp["lengths"].tolist() == [1024, 799]
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 50).cpu()
    assert torch.isfinite(out).all()
    assert float((out * (1 - inp["mask"])).abs().max()) == 0.0
    err = relerr(out, ref)
    print("N=50 T=1024: max|ref| %.4g  max|err| %.3e  rel %.2e" % (float(ref.abs().max()), float((out - ref).abs().max()), err))
    assert err <= REL


def test_reverse_diffusion_n50_t1024_mel_scale_abs(S, dev):
    """Same N and T on the mel-scale fixture (sample stays |x| < 20): the north-star's literal 1e-3 max-abs."""
    sd, inp, ref = oracle_n50("melscale")
    plan = S.Plan()
    blob = plan.pack(sd, dev)
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 50).cpu()
    print("mel-scale N=50: max|ref| %.4g  max|err| %.3e" % (float(ref.abs().max()), float((out - ref).abs().max())))
    assert 1.0 < float(ref.abs().max()) < 20
    assert float((out - ref).abs().max()) <= 1e-3


# ------------------------------------------------------------------------------------------------ per-op parity
RESNETS = ["downs.0.0", "downs.0.1", "downs.1.0", "downs.1.1", "downs.2.0", "downs.2.1", "mid_block1", "mid_block2",
           "ups.0.0", "ups.0.1", "ups.1.0", "ups.1.1"]
ATTNS = {"downs.0.2": "downs.0.1", "downs.1.2": "downs.1.1", "downs.2.2": "downs.2.1", "mid_attn": "mid_block1",
         "ups.0.2": "ups.0.1", "ups.1.2": "ups.1.1"}


def _run_with_taps(S, dev, sd, B, T, n_spks=1, seed=1234, conv_ws=False):
    inp = O.make_inputs(B, T, seed=seed, spk_dim=64 if n_spks > 1 else None)
    t = torch.linspace(0.15, 0.9, B)
    taps = {}
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t, inp.get("spk"), taps=taps)
    plan = S.Plan(n_spks=n_spks, keep_intermediates=True, conv_ws=conv_ws)
    blob = plan.pack(sd, dev)
    out = plan.estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev),
                                 inp["spk"].to(dev) if n_spks > 1 else None)
    torch.cuda.synchronize()
    hip = {k: v.detach().cpu().clone() for k, v in plan.tensors(B, T, dev).items()}
    return inp, taps, ref, hip, out.cpu()


@pytest.mark.parametrize("conv_ws", [False, True], ids=["conv_mfma", "conv_ws"])
@pytest.mark.parametrize("n_spks,B,T", [(1, 2, 64), (4, 3, 100)])
def test_every_op_output_matches_oracle_taps(S, dev, n_spks, B, T, conv_ws):
    """SURVEY section 4 'kernel' row: every Block conv (*.raw), ResnetBlock tail (*.out), attention output,
    Downsample / Upsample output and the stacked input against the oracle's taps of the same call."""
    sd = O.make_estimator_state(seed=0, n_spks=n_spks)
    inp, taps, ref, hip, out = _run_with_taps(S, dev, sd, B, T, n_spks, conv_ws=conv_ws)
    checked = 0
    worst = ("", 0.0)
    for name, want in taps.items():
        if name == "t_emb" or name.endswith(".tb") or name == "est" or name not in hip:
            continue
        got = hip[name]
        assert got.shape == want.shape, name
        e = relerr(got, want)
        if e > worst[1]:
            worst = (name, e)
        checked += 1
    # 12 resnets x (b1.raw, b2.raw, out) + 6 attention + 2 down + 2 up + final_block.raw + x0
    assert checked == 12 * 3 + 6 + 2 + 2 + 1 + 1, checked
    print("worst tap: %s rel %.2e" % worst)
    assert worst[1] <= REL, worst
    # time-bias rows (time_mlp + per-ResnetBlock projections)
    stride = hip["tb"].shape[1]
    tb = hip["tb"].view(-1, stride)[:B]
    off = 0
    for n in RESNETS:
        c = taps[n + ".tb"].shape[1]
        assert relerr(tb[:, off:off + c], taps[n + ".tb"]) <= 1e-5, n
        off += c
    assert relerr(tb[:, off:off + 64], taps["t_emb"]) <= 1e-5
    assert relerr(out, ref) <= REL


def test_local_op_parity_on_hip_inputs(S, dev):
    """Each op checked in isolation: the oracle's op applied to the HIP path's OWN input tensor, so an error cannot
    hide behind (or be blamed on) its producers.  Covers the 3x3 Block conv with both prologues, GroupNorm scale/shift,
    both ResnetBlock tails (identity and 1x1 res_conv), Downsample, Upsample (ConvTranspose) and the final conv."""
    sd = O.make_estimator_state(seed=3)
    B, T = 2, 72
    inp, taps, ref, hip, out = _run_with_taps(S, dev, sd, B, T, seed=77)
    m0 = inp["mask"].unsqueeze(1)
    masks = [m0, m0[..., ::2], m0[..., ::4]]
    worst = {}

    def chk(kind, name, got, want, tol=REL):
        e = relerr(got, want)
        worst[kind] = max(worst.get(kind, 0.0), e)
        assert e <= tol, (kind, name, e)

    def gn_apply(raw, p):
        return O.mish(F.group_norm(raw, 8, sd[p + "block.1.weight"], sd[p + "block.1.bias"], eps=1e-5))

    def resnet(name, xin, lvl):
        m, p = masks[lvl], name + "."
        chk("block1.conv", name, hip[name + ".b1.raw"],
            F.conv2d(xin * m, sd[p + "block1.block.0.weight"], sd[p + "block1.block.0.bias"], padding=1))
        b1 = hip[name + ".b1.raw"]
        h = gn_apply(b1, p + "block1.") * m + taps[name + ".tb"][:, :, None, None]
        chk("block2.conv(GN prologue)", name, hip[name + ".b2.raw"],
            F.conv2d(h * m, sd[p + "block2.block.0.weight"], sd[p + "block2.block.0.bias"], padding=1))
        b2 = hip[name + ".b2.raw"]
        h2 = gn_apply(b2, p + "block2.") * m
        if (p + "res_conv.weight") in sd:
            chk("tail(res_conv)", name, hip[name + ".out"], h2 + F.conv2d(xin * m, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"]))
        else:
            chk("tail(identity)", name, hip[name + ".out"], h2 + xin * m)
        for blk, raw in (("b1", b1), ("b2", b2)):
            C = raw.shape[1]
            g = raw.reshape(raw.shape[0], 8, -1).double()
            mean, rstd = g.mean(-1), 1.0 / torch.sqrt(g.var(-1, unbiased=False) + 1e-5)
            gamma = sd[p + "block%s.block.1.weight" % blk[1]].double()
            beta = sd[p + "block%s.block.1.bias" % blk[1]].double()
            sc = gamma[None, :] * rstd.repeat_interleave(C // 8, 1)
            sh = beta[None, :] - mean.repeat_interleave(C // 8, 1) * sc
            chk("gn.scale", name, hip[name + ".%s.sc" % blk].view(-1, C).double(), sc, 1e-5)
            chk("gn.shift", name, hip[name + ".%s.sh" % blk].view(-1, C).double(), sh, 1e-5)
        return hip[name + ".out"]

    def attn(name, xin):
        chk("attention", name, hip[name + ".out"], O.attn_residual(sd, name + ".", xin))
        return hip[name + ".out"]

    x = hip["x0"]
    hidden = []
    for lv in range(3):
        x = resnet("downs.%d.0" % lv, x, lv)
        x = resnet("downs.%d.1" % lv, x, lv)
        x = attn("downs.%d.2" % lv, x)
        hidden.append(x)
        if lv < 2:
            chk("downsample", str(lv), hip["downs.%d.3.out" % lv],
                F.conv2d(x * masks[lv], sd["downs.%d.3.conv.weight" % lv], sd["downs.%d.3.conv.bias" % lv], stride=2, padding=1))
            x = hip["downs.%d.3.out" % lv]
    x = resnet("mid_block1", x, 2)
    x = attn("mid_attn", x)
    x = resnet("mid_block2", x, 2)
    for u in range(2):
        lv = 2 - u
        x = torch.cat((x, hidden.pop()), 1)
        x = resnet("ups.%d.0" % u, x, lv)
        x = resnet("ups.%d.1" % u, x, lv)
        x = attn("ups.%d.2" % u, x)
        chk("upsample", str(u), hip["ups.%d.3.out" % u],
            F.conv_transpose2d(x * masks[lv], sd["ups.%d.3.conv.weight" % u], sd["ups.%d.3.conv.bias" % u], stride=2, padding=1))
        x = hip["ups.%d.3.out" % u]
    chk("final_block.conv", "", hip["final_block.raw"],
        F.conv2d(x * m0, sd["final_block.block.0.weight"], sd["final_block.block.0.bias"], padding=1))
    fb = gn_apply(hip["final_block.raw"], "final_block.") * m0
    est = (F.conv2d(fb * m0, sd["final_conv.weight"], sd["final_conv.bias"]) * m0).squeeze(1)
    chk("final_conv", "", out, est)
    print("local per-op worst rel errors: " + ", ".join("%s %.1e" % kv for kv in sorted(worst.items())))


def test_linear_attention_with_unit_rezero_gain(S, dev):
    """Rezero.g = 1.0: the attention branch at full strength, single call (a single call is stable; only long
    trajectories blow up).  Asserted on the branch itself, (A.out - x), so the residual cannot mask an error, both
    end-to-end against the oracle's taps and locally on the HIP path's own input."""
    sd = O.make_estimator_state(seed=5, rezero_g=1.0)
    B, T = 2, 64
    inp, taps, ref, hip, out = _run_with_taps(S, dev, sd, B, T, seed=31)
    for a, prev in ATTNS.items():
        xin_h, xin_r = hip[prev + ".out"], taps[prev + ".out"]
        br_ref = taps[a + ".out"] - xin_r
        br_hip = hip[a + ".out"] - xin_h
        br_loc = O.linear_attention(sd, a + ".fn.fn.", xin_h) * sd[a + ".fn.g"]
        scale = float(br_ref.abs().max())
        assert scale > 1e-3, (a, scale)                     # the branch is not negligible
        e_loc = float((br_hip - br_loc).abs().max()) / float(br_loc.abs().max())
        e_end = float((br_hip - br_ref).abs().max()) / scale
        print("%-10s branch max %.3g (x max %.3g)  local rel %.2e  end-to-end rel %.2e" % (a, scale, float(xin_r.abs().max()), e_loc, e_end))
        assert e_loc <= 2e-4, (a, e_loc)
        assert e_end <= 1e-3, (a, e_end)
    assert relerr(out, ref) <= 5e-4


# ------------------------------------------------------------------------------------------------ config 3
def _gradtts(n_spks, dev, seed=2):
    M = importlib.import_module("speech-backbones_amd.model")
    torch.manual_seed(0)
    model = M.GradTTS(149, n_spks, 64, 192, 768, 256, 2, 6, 3, 0.1, 4, 80, 64, 0.05, 20.0, 1000)
    sd = O.make_estimator_state(seed=seed, n_spks=n_spks)
    model.decoder.estimator.load_state_dict(sd, strict=True)
    with torch.no_grad():
        model.encoder.proj_w.proj.bias.fill_(1.2)
    return model.to(dev).eval(), sd


def _tts_reference(model, sd, x, xl, spk_ids, n_steps, length_scale, seed, dev):
    """Host-side restatement of GradTTS.forward around the oracle decoder (teacher-forced durations)."""
    with torch.no_grad():
        emb = model.spk_emb(spk_ids.to(dev)).cpu() if spk_ids is not None else None
        mu_x, logw, x_mask = model.encoder(x.to(dev), xl.to(dev), None)
    w_ceil = (torch.ceil(torch.exp(logw) * x_mask) * length_scale).cpu()
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_max = int(y_lengths.max())
    y_max_ = O.fix_len_compatibility(y_max)
    y_mask = O.sequence_mask(y_lengths, y_max_).unsqueeze(1).float()
    path = O.generate_path(w_ceil.squeeze(1), (x_mask.cpu().unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1))
    mu_y = torch.matmul(path.transpose(1, 2), mu_x.cpu().transpose(1, 2)).transpose(1, 2)
    torch.manual_seed(seed)
    tmpl = torch.empty(mu_y.shape[0], mu_y.shape[2], mu_y.shape[1], device=dev).transpose(1, 2)
    z = mu_y + torch.randn_like(tmpl).cpu() / 1.5
    return O.reverse_diffusion(sd, z, y_mask, mu_y, n_steps, spk=emb)[:, :, :y_max], y_max


def test_gradtts_multispeaker_forward_drop_in(S, dev):
    """GradTTS(n_spks=247).forward(spk=ids) (tts.py:43-44,70-72; diffusion.py:139-141,175-176,183-185) through the
    module path: speaker embedding -> spk_mlp -> third input channel, fp32-grade (bf16x3) precision."""
    model, sd = _gradtts(247, dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randint(0, 149, (3, 19), generator=g)
    xl = torch.tensor([19, 11, 16])
    spk = torch.tensor([0, 246, 101])
    torch.manual_seed(77)
    enc, dec_out, attn = model(x.to(dev), xl.to(dev), n_timesteps=4, temperature=1.5, spk=spk.to(dev), length_scale=0.91)
    ref, y_max = _tts_reference(model, sd, x, xl, spk, 4, 0.91, 77, dev)
    assert dec_out.shape == ref.shape
    assert relerr(dec_out.cpu(), ref) <= REL
    # a different speaker changes the sample (the conditioning path is live)
    torch.manual_seed(77)
    other = model(x.to(dev), xl.to(dev), n_timesteps=4, temperature=1.5, spk=torch.tensor([5, 5, 5]).to(dev), length_scale=0.91)[1]
    assert float((other - dec_out).abs().max()) > 1e-3


def test_config3_bf16_multispeaker_teacher_forced_n100(S, dev):
    """BASELINE config 3: 247 speakers, plain-bf16 contractions, N=100.  The untrained reverse ODE amplifies any
    perturbation e^5-fold, so bf16 is judged teacher-forced: the oracle's own x_t of each of the 100
    steps is fed to the HIP estimator through Diffusion.estimator(..., spk=...).  Tolerance: 2.5e-2 * max|ref| per call
    (bf16 has 8 mantissa bits; measured ~8e-3), and the bf16x3 mode on the same inputs must stay <= 1e-4."""
    M = importlib.import_module("speech-backbones_amd.model.diffusion")
    sd = O.make_estimator_state(seed=7, n_spks=247)
    dec = M.Diffusion(80, 64, 247, 64, 0.05, 20.0, 1000)
    dec.estimator.load_state_dict(sd, strict=True)
    dec = dec.to(dev).eval()
    inp = O.make_inputs(2, 48, seed=5, spk_dim=64)
    traj = []
    O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 100, spk=inp["spk"], trajectory=traj)
    assert len(traj) == 100
    m, mu, spk = inp["mask"].to(dev), inp["mu"].to(dev), inp["spk"].to(dev)
    h = 1.0 / 100
    worst = {}
    for prec, every in (("bf16", 1), ("bf16_store", 1), ("bf16x3", 10)):
        dec.estimator.set_precision(prec)
        w = 0.0
        for i in range(0, 100, every):
            xt, est = traj[i]
            t = torch.full((2,), np.float32(1.0 - (i + 0.5) * h))
            with torch.no_grad():
                got = dec.estimator(xt.to(dev), m, mu, t.to(dev), spk).cpu()
            w = max(w, relerr(got, est))
        worst[prec] = w
    print("teacher-forced N=100, 247 speakers: worst rel err bf16 %.2e, bf16_store %.2e, bf16x3 %.2e" %
          (worst["bf16"], worst["bf16_store"], worst["bf16x3"]))
    assert worst["bf16x3"] <= REL
    assert worst["bf16x3"] < worst["bf16"] <= REL_BF16
    assert worst["bf16_store"] <= REL_BF16_STORE


def test_config3_bf16_gradtts_forward_with_spk(S, dev):
    """config 3 through the outermost entry point: GradTTS(n_spks=247).forward(spk=...) with the decoder in bf16 mode.
    Free-running N=10 on the mel-scale-stabilised score (final layer x0.1) so that bf16's per-call error is not amplified
    beyond the stated bound: max|err| <= 2.5e-2 * max|ref|."""
    model, sd = _gradtts(247, dev, seed=9)
    sd = dict(sd)
    sd["final_conv.weight"] = sd["final_conv.weight"] * 0.1
    sd["final_conv.bias"] = sd["final_conv.bias"] * 0.1
    model.decoder.estimator.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(8)
    x = torch.randint(0, 149, (2, 21), generator=g)
    xl = torch.tensor([21, 14])
    spk = torch.tensor([17, 200])
    ref, _ = _tts_reference(model, sd, x, xl, spk, 10, 1.0, 3, dev)
    for prec, tol in (("bf16", REL_BF16), ("bf16_store", REL_BF16_STORE)):
        model.decoder.estimator.set_precision(prec)
        torch.manual_seed(3)
        dec_out = model(x.to(dev), xl.to(dev), n_timesteps=10, temperature=1.5, spk=spk.to(dev))[1]
        e = relerr(dec_out.cpu(), ref)
        print("GradTTS(247 spk, %s) N=10 free-running rel err %.2e" % (prec, e))
        assert e <= tol


def test_config3_bf16_store_n100_free_running_mel_scale(S, dev):
    """BASELINE config 3 at its own N: 247 speakers, N = 100, bf16 contractions AND bf16 activation storage, FREE-RUNNING on the
    mel-scale fixture (final layer x 0.1: the sample stays |x| < 40, the regime a trained score keeps it in).  Stated bound:
    max|err| <= 0.6 at a sample scale of 23.5, i.e. 2.6 % (measured 0.33 = 1.4 %, printed); bf16x3 on the same run <= 2e-3
    (measured 2.7e-4)."""
    sd = dict(O.make_estimator_state(seed=7, n_spks=247))
    sd["final_conv.weight"] = sd["final_conv.weight"] * 0.1
    sd["final_conv.bias"] = sd["final_conv.bias"] * 0.1
    inp = O.make_inputs(1, 256, seed=21, temperature=150.0, ragged=False, spk_dim=64)
    ref = O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 100, spk=inp["spk"])
    assert 1.0 < float(ref.abs().max()) < 40
    errs = {}
    for prec in ("bf16_store", "bf16x3"):
        plan = S.Plan(n_spks=247, precision={"bf16_store": S.PREC_BF16_STORE, "bf16x3": S.PREC_BF16X3}[prec])
        blob = plan.pack(sd, dev)
        out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 100, spk=inp["spk"].to(dev)).cpu()
        assert torch.isfinite(out).all()
        errs[prec] = float((out - ref).abs().max())
    print("config 3 free-running N=100 mel scale: max|ref| %.3g  max|err| bf16_store %.3e  bf16x3 %.3e" %
          (float(ref.abs().max()), errs["bf16_store"], errs["bf16x3"]))
    assert errs["bf16x3"] <= 2e-3
    assert errs["bf16_store"] <= 0.6


# ------------------------------------------------------------------------------------------------ config 4
def test_vc_dim256_t1024_single_call(S, dev):
    """BASELINE config 4 shape: DiffVC decoder dim 256 (117.8 M parameters) on an 80x1024 utterance, one estimator
    call (DiffVC/model/diffusion.py:61-106) -- ~2 TFLOP on the CPU oracle."""
    sd = V.make_state(dim_base=256, seed=1)
    plan = S.Plan(dim=256, arch=1)
    blob = plan.pack(sd, dev)
    inp = V.make_inputs(1, 1024, 256, seed=3, ragged=False)
    t = torch.tensor([0.7])
    xt_ref = torch.stack([V.compute_diffused_mean(inp["ref"], inp["ref_mask"], inp["mean_ref"], 0.7)], 1)
    ref = V.estimator_forward(sd, inp["z"], inp["mask"], inp["mean"], xt_ref, inp["ref_mask"], inp["c"], t)
    out = plan.vc_estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mean"].to(dev), xt_ref.to(dev),
                                    inp["ref_mask"].to(dev), inp["c"].to(dev), t.to(dev)).cpu()
    e = relerr(out, ref)
    print("DiffVC dim256 T=1024: rel err %.2e" % e)
    assert e <= REL


# ------------------------------------------------------------------------------------------------ hipGraph replay
def test_graph_replay_is_bit_identical_to_eager(S, dev):
    """gtts_plan_set_graph: the captured-and-replayed sampler call (B=1, the inference.py regime) returns exactly the
    eager result, on the capture call, on replays, and after the inputs changed in place (same addresses)."""
    sd = O.make_estimator_state(seed=0)
    eager = S.Plan()
    blob = eager.pack(sd, dev)
    gp = S.Plan()
    gp.set_graph(True)
    for seed in (1, 2, 3):
        inp = O.make_inputs(1, 64, seed=seed, ragged=False)
        z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
        want = eager.reverse_diffusion(blob, z, m, mu, 5)
        got = gp.reverse_diffusion(blob, z, m, mu, 5)
        assert torch.equal(got, want), seed
    inp = O.make_inputs(3, 32, seed=9)                     # another shape: a second graph, sub-batch streams inside it
    z, m, mu = inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev)
    for _ in range(2):
        assert torch.equal(gp.reverse_diffusion(blob, z, m, mu, 3), eager.reverse_diffusion(blob, z, m, mu, 3))
    ref = O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["mu"], 3)
    assert relerr(gp.reverse_diffusion(blob, z, m, mu, 3).cpu(), ref) <= REL
    gp.set_graph(False)
    assert torch.equal(gp.reverse_diffusion(blob, z, m, mu, 3), eager.reverse_diffusion(blob, z, m, mu, 3))


# ------------------------------------------------------------------------------------------------ f2: MAS score matrix
def test_log_prior_kernel_and_mas_on_device(S, dev):
    """gtts_log_prior (tts.py:130-139) at training shapes: the matrix itself <= 1e-5 * max|ref| against the reference's
    three-matmul expression, and -- bit-exactness being required of the PATH given identical scores -- the GPU MAS on the
    GPU scores equals the plain-C oracle MAS fed the same GPU scores."""
    import math
    from oracle import mas as MAS
    g = torch.Generator().manual_seed(12)
    B, Fm, tx, T = 4, 80, 57, 236
    mu_x = torch.randn(B, Fm, tx, generator=g)
    y = torch.randn(B, Fm, T, generator=g) * 1.3
    factor = -0.5 * torch.ones_like(mu_x)
    ref = (torch.matmul(factor.transpose(1, 2), y ** 2) - torch.matmul(2.0 * (factor * mu_x).transpose(1, 2), y)
           + torch.sum(factor * mu_x ** 2, 1).unsqueeze(-1) - 0.5 * math.log(2 * math.pi) * Fm)
    got = S._lib.log_prior(mu_x.to(dev), y.to(dev))
    assert got.shape == (B, tx, T)
    assert relerr(got.cpu(), ref) <= 1e-5
    xl = torch.tensor([57, 40, 13, 57])
    yl = torch.tensor([236, 200, 90, 150])
    mask = (O.sequence_mask(xl, tx).unsqueeze(-1) * O.sequence_mask(yl, T).unsqueeze(1)).float()
    path = S.mas_maximum_path(got, mask.to(dev)).cpu()
    assert torch.equal(path, MAS.maximum_path_port(got.cpu(), mask))


def test_gradtts_compute_loss_on_gpu(S, dev):
    """GradTTS.compute_loss on the GPU (encoder -> gtts_log_prior -> gtts_mas_maximum_path -> losses) against the same
    module on the CPU: duration and prior losses are deterministic given the weights and must agree; the diffusion loss
    draws device noise and is only checked for finiteness and a gradient on every decoder parameter."""
    import copy
    M = importlib.import_module("speech-backbones_amd.model")
    torch.manual_seed(0)
    model = M.GradTTS(149, 1, 64, 192, 768, 256, 2, 6, 3, 0.1, 4, 80, 64, 0.05, 20.0, 1000).train()
    for mod in model.modules():                   # dropout off: CPU and GPU draw different masks
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    cpu = copy.deepcopy(model)
    model = model.to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 149, (3, 41), generator=g)
    xl = torch.tensor([41, 30, 12])
    y = torch.randn(3, 80, 120, generator=g)
    yl = torch.tensor([120, 96, 60])
    a = model.compute_loss(x.to(dev), xl.to(dev), y.to(dev), yl.to(dev))
    b = cpu.compute_loss(x, xl, y, yl)
    assert abs(float(a[0]) - float(b[0])) <= 1e-4 * max(1.0, abs(float(b[0])))
    assert abs(float(a[1]) - float(b[1])) <= 1e-4 * max(1.0, abs(float(b[1])))
    assert torch.isfinite(a[2])
    sum(a).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.decoder.parameters())
