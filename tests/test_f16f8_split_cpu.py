"""The arithmetic of GTTS_PREC_F16F8 restated on the CPU (speech-backbones_amd/csrc/common.h F8_S / F8_D, pack.hip, conv_ws.hip; the op is
the Block convolution of Grad-TTS/model/diffusion.py:49-58): w.x = w_h.x_h on an fp16 MFMA + BOTH cross terms w.x_l + w_l.x in one fp8
(e4m3) MFMA with K block 0 = q8(w).q8(x_l 2^S), K block 1 = q8(w_l 2^(S+D)).q8(x 2^-D), accumulators holding 2^S x the sum.  Operand
roundings only (products and sums in fp64), on a K = 1152 reduction like the 128-channel 3x3 layers: the split must stay in the error
class of bf16x3 (the device measures 1.26e-5 against 4.2e-6, profiles/r05_f8_probe2.txt), the cross terms must matter, and an operand
beyond the fp8 range must cost only its own cross term (saturation, never NaN)."""
import torch

S, D = 10, 4


def f16(x):
    return x.to(torch.float16).double()


def bf16(x):
    return x.to(torch.bfloat16).double()


def q8(x):
    return x.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).double()     # v_med3_f32 + v_cvt_pk_fp8_f32


def f16f8(w, x):
    wh, xh = f16(w), f16(x)
    wl, xl = w - wh, x - xh
    acc = (wh * 2.0 ** S) @ xh + q8(w) @ q8(xl * 2.0 ** S) + q8(wl * 2.0 ** (S + D)) @ q8(x * 2.0 ** -D)
    return acc * 2.0 ** -S


def bf16x3(w, x):
    wh, xh = bf16(w), bf16(x)
    wl, xl = bf16(w - wh), bf16(x - xh)
    return wl @ xh + wh @ xh + wh @ xl


def data(scale_x=1.0):
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(128, 1152, generator=g) * 0.05).double()
    x = (torch.randn(1152, 512, generator=g) * scale_x).double()
    x = x * (torch.rand(1152, 1, generator=g) * 4).double()                 # channels of different magnitude, like post-Mish activations
    return w.float().double(), x.float().double()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_split_is_fp32_grade_and_needs_its_cross_terms():
    w, x = data()
    ref = w @ x
    e8, e3 = rel(f16f8(w, x), ref), rel(bf16x3(w, x), ref)
    e_hi = rel(f16(w) @ f16(x), ref)
    print("f16f8 %.2e  bf16x3 %.2e  fp16 hi*hi alone %.2e" % (e8, e3, e_hi))
    assert e3 < 1e-5 and e8 < 4e-5 and e8 < 8 * e3
    assert e_hi > 10 * e8                                                    # the fp8 MFMA carries what a single fp16 pass drops


def test_weights_scaled_by_2s_stay_exact_in_fp16():
    w, _ = data()
    wh = f16(w)
    assert torch.equal(f16(wh * 2.0 ** S), wh * 2.0 ** S)                    # |w| < 63: pack.hip stores fp16(w 2^S) without a second rounding
    assert float((wh * 2.0 ** S).abs().max()) < 65504


def test_saturated_operands_degrade_to_fp16_grade_not_nan():
    w, x = data(scale_x=400.0)                                               # |x| up to ~6000: q8(x_l 2^S) saturates for the large entries
    assert float((x - f16(x)).abs().max() * 2.0 ** S) > 448
    ref = w @ x
    out = f16f8(w, x)
    assert torch.isfinite(out).all()
    e8, e_hi = rel(out, ref), rel(f16(w) @ f16(x), ref)
    print("saturating inputs: f16f8 %.2e  fp16 hi*hi alone %.2e" % (e8, e_hi))
    assert e8 <= e_hi * 1.05 and e8 < 2e-3


def test_weight_range_rule_of_the_packer():
    """pack.hip's range check restated (round-5 review 3a): fp16(w 2^S) is finite exactly for |w| <= 65504 / 2^S = 63.97; beyond it the
    packer SATURATES the hi part (never inf) and counts the weight, and gtts_pack_weights turns a non-zero count into GTTS_E_RANGE."""
    wmax = 65504.0 / 2.0 ** S
    inside = torch.tensor([0.0, 1.0, -63.9, wmax, -wmax])
    # (fp16 rounds to nearest: values up to 65519.99 still round DOWN to 65504 -- the packer's bound is the representable maximum itself,
    # so everything it accepts is exactly in range and everything above is refused, including the few that would round down)
    assert torch.isfinite((inside * 2.0 ** S).to(torch.float16)).all()
    outside = torch.tensor([64.0, -70.0, 1e4, float("inf")])
    assert not (outside.abs() <= wmax).any()
    assert torch.isinf((torch.tensor([64.0, -70.0, 1e4]) * 2.0 ** S).to(torch.float16)).all()       # what the old packer wrote
    sat = outside.clamp(-wmax, wmax)
    assert torch.isfinite((sat * 2.0 ** S).to(torch.float16)).all()                                   # what the packer writes now
    # the static side: the device check and the error path exist where the header says they do
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pack = open(os.path.join(root, "speech-backbones_amd", "csrc", "pack.hip")).read()
    plan = open(os.path.join(root, "speech-backbones_amd", "csrc", "plan.hip")).read()
    hdr = open(os.path.join(root, "include", "gradtts_abi.h")).read()
    assert "WMAX = 65504.0f / (float)(1 << F8_S)" in pack and "atomicAdd(status + 0, 1u)" in pack
    assert "return fail(GTTS_E_RANGE" in plan and "hipStreamSynchronize(st)" in plan
    assert "GTTS_E_RANGE = -7" in hdr and "gtts_workspace_status" in hdr


def test_activation_range_threshold():
    """common.h F8_ACT_LIMIT = 1024: the smallest |x| whose fp16 residual can leave the fp8 operand's range (|x_l| 2^S > 448)."""
    x = torch.linspace(512.0, 1023.99, 20001, dtype=torch.float64)
    assert float(((x - f16(x)).abs() * 2.0 ** S).max()) <= 448.0
    x = torch.linspace(1024.0, 2047.0, 20001, dtype=torch.float64)
    assert float(((x - f16(x)).abs() * 2.0 ** S).max()) > 448.0
