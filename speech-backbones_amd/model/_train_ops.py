"""Autograd composition of the score U-Net -- TRAINING ONLY (the sampling path under torch.no_grad never comes here).

Training (Diffusion.compute_loss -> loss_t -> estimator with autograd, Grad-TTS/model/diffusion.py:281-294) needs gradients
w.r.t. the same nn.Parameters.  On HIP tensors every tensor-sized op of the network runs on the hand-written kernels of
csrc/train*.hip behind torch.autograd.Function wrappers:
  MaskedConv3x3        Block's 3x3 convolution (forward / data gradient on the inference MFMA kernel, weight gradient as a
                       wave-specialised MFMA reduction over pixels); the up path's torch.cat is read in place (two sources)
  GnMishMask           GroupNorm + Mish + mask (+ ResnetBlock's time term), forward and backward fused
  MaskedConv1x1        res_conv, to_qkv, to_out (forward / data gradient on the CONV_P1 kernel, MFMA weight gradient)
  LinearAttentionCore  softmax over pixels, context, output (forward and backward)
  RezeroResidual, MaskedResidualAdd, the plain residual add, FinalConv (64 -> 1 with both masks), ScoreLoss
  ResampleConv         Downsample / Upsample forward and all their gradients on the HIP kernels (Downsample's data gradient is an
                       Upsample call, Upsample's gradients are 3x3 stride-1 operations over the four stride-2 phases of dy)
The [B, dim] time / speaker MLPs stay stock torch ops.  On CPU tensors (tests) everything is stock torch.
"""
import math

import torch
import torch.nn.functional as F

from ._backend import backend


class MaskedConv3x3(torch.autograd.Function):
    """y = Conv2d_3x3(cat(x, x1) * mask) + bias with all three gradients on the HIP kernels (x1 None: one source)."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias, x1=None):
        be = backend()
        cols = mask.reshape(mask.shape[0], mask.shape[-1])          # [B,1,1,W] -> [B,W]
        ctx.save_for_backward(x, cols, weight, x1)
        return be.conv3x3_masked(x, cols, weight, bias, x1)

    @staticmethod
    def backward(ctx, dy):
        be = backend()
        x, cols, weight, x1 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dx1 = dw = db = None
        need1 = x1 is not None and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[0] or need1:
            c0 = x.shape[1]
            if x1 is None:
                dx = be.conv3x3_dgrad(dy, weight, cols)             # (the mask rides in the convolution's epilogue)
            else:                                                   # mask and split in one pass each (contiguous results)
                full = be.conv3x3_dgrad(dy, weight)
                if ctx.needs_input_grad[0]:
                    dx = be.add_masked(None, full, cols, channels=(0, c0))
                if need1:
                    dx1 = be.add_masked(None, full, cols, channels=(c0, full.shape[1]))
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:      # (one kernel produces both)
            dw, db = be.conv3x3_wgrad(x, cols, dy, x1)
        return dx, None, (dw if ctx.needs_input_grad[2] else None), (db if ctx.needs_input_grad[3] else None), dx1


class MaskedConv1x1(torch.autograd.Function):
    """y = Conv2d_1x1(x * mask) + bias (res_conv, to_qkv, to_out: diffusion.py:70,87-88; mask and bias optional) with all three
    gradients on the HIP kernels.  A 1x1 convolution does not mix columns, so the mask of the data gradient is applied to dy
    by the same kernel prologue."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias):
        be = backend()
        cols = None if mask is None else mask.reshape(mask.shape[0], mask.shape[-1])
        ctx.save_for_backward(x, cols, weight)
        ctx.has_bias = bias is not None
        return be.conv1x1_masked(x, cols, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        be = backend()
        x, cols, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = be.conv1x1_dgrad(dy, weight, cols)
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw, db = be.conv1x1_wgrad(x, cols, dy, want_bias=ctx.has_bias)
        return dx, None, (dw if ctx.needs_input_grad[2] else None), (db if ctx.has_bias and ctx.needs_input_grad[3] else None)


class LinearAttentionCore(torch.autograd.Function):
    """softmax over pixels of k, context = k~ v^T, out = context^T q (LinearAttention.forward between its two 1x1 convolutions,
    diffusion.py:90-100) on to_qkv's output [B, 384, H, W]; forward and backward on csrc/train_attn.hip."""

    @staticmethod
    def forward(ctx, qkv):
        out, c, stat = backend().attn_train_forward(qkv)
        ctx.save_for_backward(qkv, c, stat)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, c, stat = ctx.saved_tensors
        return backend().attn_train_backward(qkv, dout.contiguous(), c, stat)


class RezeroResidual(torch.autograd.Function):
    """f * g + x (Residual(Rezero(fn)), diffusion.py:40-46,103-108) with d f = dy * g, d g = sum(dy * f), d x = dy."""

    @staticmethod
    def forward(ctx, f, g, x):
        ctx.save_for_backward(f, g)
        return backend().rezero_forward(f, g, x)

    @staticmethod
    def backward(ctx, dy):
        f, g = ctx.saved_tensors
        dy = dy.contiguous()
        df, dg = backend().rezero_backward(dy, f, g)
        return df, dg, dy


class GnMishMask(torch.autograd.Function):
    """Mish(GroupNorm(y)) * mask (Block.forward, diffusion.py:53-58) [+ tb[:, :, None, None]: ResnetBlock's time term,
    diffusion.py:75-76] with forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, y, mask, gamma, beta, groups, eps, tb=None):
        be = backend()
        cols = mask.reshape(mask.shape[0], mask.shape[-1])          # [B,1,1,W] -> [B,W]
        out, stats = be.gn_mish_forward(y, gamma, beta, cols, groups, eps, tb)
        ctx.save_for_backward(y, cols, gamma, beta, stats)
        ctx.groups = groups
        ctx.has_tb = tb is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        y, cols, gamma, beta, stats = ctx.saved_tensors
        r = backend().gn_mish_backward(dout.contiguous(), y, gamma, beta, cols, stats, ctx.groups, want_dtb=ctx.has_tb)
        return r[0], None, r[1], r[2], None, None, (r[3] if ctx.has_tb else None)


class InstNormGlu(torch.autograd.Function):
    """InstanceNorm2d(affine) -> GLU(dim=1) on a convolution output [B, 2C, H, W] (DiffVC RefBlock, DiffVC/model/modules.py:128-157) with
    forward and backward on csrc/train_inglu.hip."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        out, stats = backend().in_glu_forward(y, gamma, beta, eps)
        ctx.save_for_backward(y, gamma, beta, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, stats = ctx.saved_tensors
        dy, dg, db = backend().in_glu_backward(dout.contiguous(), y, gamma, beta, stats)
        return dy, dg, db, None


class MaskedResidualAdd(torch.autograd.Function):
    """h + v * mask (ResnetBlock with an identity res_conv, diffusion.py:77-78; mask None: h + v)."""

    @staticmethod
    def forward(ctx, h, v, mask):
        cols = None if mask is None else mask.reshape(mask.shape[0], mask.shape[-1])
        ctx.save_for_backward(cols)
        return backend().add_masked(h, v, cols)

    @staticmethod
    def backward(ctx, dout):
        (cols,) = ctx.saved_tensors
        dout = dout.contiguous()
        dv = dout if cols is None else (backend().add_masked(None, dout, cols) if ctx.needs_input_grad[1] else None)
        return dout, dv, None


class FinalConv(torch.autograd.Function):
    """(final_conv(x * mask)) * mask for the 64 -> 1 convolution (diffusion.py:175-176)."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias):
        cols = mask.reshape(mask.shape[0], mask.shape[-1])
        ctx.save_for_backward(x, cols, weight)
        return backend().final_conv_forward(x, weight, bias, cols)

    @staticmethod
    def backward(ctx, dout):
        x, cols, weight = ctx.saved_tensors
        dx, dw, db = backend().final_conv_backward(x, weight, cols, dout.contiguous())
        return dx, None, dw, db


class ResampleConv(torch.autograd.Function):
    """Downsample / Upsample of x * mask (diffusion.py:19-34,158,171): forward on the inference kernels; Downsample's data
    gradient is an Upsample call with the zero-padded kernel and its weight gradient the stride-1 MFMA reduction against the
    zero-inserted dy; Upsample's gradients are the 3x3 convolution / weight gradient over the four stride-2 phases of dy."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias, up):
        cols = mask.reshape(mask.shape[0], mask.shape[-1])
        ctx.save_for_backward(x, cols, weight)
        ctx.up = bool(up)
        ctx.bias_n = int(bias.shape[0])
        return backend().conv_resample(x, cols, weight, bias, up)

    @staticmethod
    def backward(ctx, dy):
        be = backend()
        x, cols, weight = ctx.saved_tensors
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        if not ctx.up:
            # stride 2: the weight gradient is the stride-1 one against the zero-inserted dy (MFMA reduction of train_wgrad.hip)
            gw, gb = be.conv3x3_wgrad(x, cols, be.zero_insert2(dy))
            gi = None
            if need_dx:
                ones = be._const(dy.device, "ones", int(dy.shape[0]), int(dy.shape[3]))
                gi = be.conv_resample(dy, ones, weight, None, True, dgrad_of_down=True)
        else:
            # ConvTranspose2d 4x4 stride 2 pad 1: output row oy = 2 iy - 1 + ky.  Over the four stride-2 phases of dy (even / odd
            # rows x columns, stacked as channels) both gradients are stride-1 3x3 operations: odd rows meet taps ky = 0, 2 at
            # offsets -1, 0; even rows taps ky = 1, 3 at offsets 0, +1 (the same for columns).
            P = be.space_to_depth2(dy)                              # [B, 4 cout, h, w], block (pr * 2 + pc), pr = 1: even rows
            ci, co = int(weight.shape[0]), int(weight.shape[1])
            tap = _up_tap_index(dy.device)                          # [2 (phase), 3 (ky')] -> ky, 4 = no tap
            ones = be._const(dy.device, "ones", int(dy.shape[0]), int(x.shape[3]))
            gi = None
            if need_dx:
                wz = torch.cat((weight.reshape(ci, co, 4, 4), weight.new_zeros(ci, co, 1, 4)), dim=2)
                wz = torch.cat((wz, wz.new_zeros(ci, co, 5, 1)), dim=3)                        # [ci, co, 5, 5], index 4 = zero
                wp = wz[:, :, tap[:, None, :, None], tap[None, :, None, :]]                    # [ci, co, pr, pc, ky', kx']
                wp = wp.permute(0, 2, 3, 1, 4, 5).reshape(ci, 4 * co, 3, 3).contiguous()
                gi = be.conv3x3_masked(P, ones, wp, be._const(dy.device, "zeros", ci))
            xm = be.add_masked(None, x, cols)
            d6, _ = be.conv3x3_wgrad(P, ones, xm)                   # [ci][4 co][3][3]: "dy" role = x * mask, "x" role = the phases
            d6 = d6.reshape(ci, 2, 2, co, 3, 3)
            pr, kp = _up_tap_inverse(dy.device)                     # ky -> (phase, ky')
            gw = d6[:, pr[:, None], pr[None, :], :, kp[:, None], kp[None, :]].permute(2, 3, 0, 1).contiguous()
            gb = dy.sum((0, 2, 3))
        dx = be.add_masked(None, gi, cols) if need_dx else None
        return dx, None, gw, gb, None


_UP_IDX = {}


def _up_tap_index(device):
    """[phase (0 odd rows, 1 even rows)][ky' of the 3x3 view] -> ky of the 4x4 kernel, 4 where the phase has no tap."""
    key = (str(device), "fwd")
    if key not in _UP_IDX:
        _UP_IDX[key] = torch.tensor([[0, 2, 4], [4, 1, 3]], dtype=torch.long, device=device)
    return _UP_IDX[key]


def _up_tap_inverse(device):
    """ky of the 4x4 kernel -> (phase, ky' of the 3x3 view)."""
    key = (str(device), "inv")
    if key not in _UP_IDX:
        _UP_IDX[key] = (torch.tensor([0, 1, 0, 1], dtype=torch.long, device=device), torch.tensor([0, 1, 1, 2], dtype=torch.long, device=device))
    return _UP_IDX[key]


class ScoreLoss(torch.autograd.Function):
    """sum((eps * sqrt(1 - e^{-cum}) + z)^2) / denom with the gradient produced in the same pass (diffusion.py:285-287)."""

    @staticmethod
    def forward(ctx, eps, z, t, beta_min, beta_max, inv_denom):
        loss, g = backend().score_loss(eps, z, t, beta_min, beta_max, inv_denom, want_grad=True)
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (g,) = ctx.saved_tensors
        return g * dloss, None, None, None, None, None


FORCE_TORCH = False      # measurement switch (bench.py train_step): route every op to stock PyTorch-ROCm kernels

# How many tensor-sized ops of the training path ran on the gtts:: kernels and how many fell back to stock torch ops because a gate
# (shape, dtype, device, channel multiple) failed.  A fallback is correct but slow, and silent otherwise: callers / tests read
# `op_counts()` after a step (tests/test_gpu_training.py asserts (n, 0) on the reference's shapes).
_COUNTS = {"hip_ops": 0, "torch_fallback_ops": 0}


def reset_op_counts():
    _COUNTS["hip_ops"] = 0
    _COUNTS["torch_fallback_ops"] = 0


def op_counts():
    return _COUNTS["hip_ops"], _COUNTS["torch_fallback_ops"]


def _count(hip):
    if not FORCE_TORCH:
        _COUNTS["hip_ops" if hip else "torch_fallback_ops"] += 1


def _hip_conv_ok(v, conv):
    # (shape: the kernels address one call's tensors with 32-bit byte offsets -- oversize batches / crops take the torch path
    # up front instead of failing inside loss.backward())
    return (not FORCE_TORCH and v.is_cuda and v.dtype == torch.float32 and conv.kernel_size == (3, 3) and
            backend().conv3x3_supported(conv.in_channels, conv.out_channels, need_dgrad=v.requires_grad,
                                        shape=(v.shape[0], v.shape[2], v.shape[3])))


def _conv1x1(v, m, conv):
    """conv(v * m) for a 1x1 nn.Conv2d (m None: no mask)."""
    if (not FORCE_TORCH and v.is_cuda and v.dtype == torch.float32 and conv.kernel_size == (1, 1) and
            backend().conv1x1_supported(conv.in_channels, conv.out_channels, need_dgrad=v.requires_grad,
                                        shape=(v.shape[0], v.shape[2], v.shape[3]))):
        _count(True)
        return MaskedConv1x1.apply(v.contiguous(), m, conv.weight, conv.bias)
    _count(False)
    return F.conv2d(v if m is None else v * m, conv.weight, conv.bias)


def _mish(v):
    return v * torch.tanh(F.softplus(v))


def _hip(v):
    return not FORCE_TORCH and v.is_cuda and v.dtype == torch.float32


def _conv_gn_mish(blk, v, m, tb=None, v1=None):
    """Block.forward [+ the time term tb[:, :, None, None]] on v (or on the concatenation cat(v, v1), read in place)."""
    conv, norm = blk.block[0], blk.block[1]
    if v1 is not None and not (_hip_conv_ok(v, conv) and v.shape[1] % 64 == 0):
        v, v1 = torch.cat((v, v1), dim=1), None
    if _hip_conv_ok(v, conv):
        _count(True)
        y = MaskedConv3x3.apply(v.contiguous(), m, conv.weight, conv.bias, None if v1 is None else v1.contiguous())
    else:
        _count(False)
        y = F.conv2d(v * m, conv.weight, conv.bias, padding=1)
    if _hip(y) and y.dim() == 4 and y.shape[1] % norm.num_groups == 0:
        _count(True)
        return GnMishMask.apply(y.contiguous(), m, norm.weight, norm.bias, norm.num_groups, norm.eps,
                                None if tb is None else tb.contiguous())
    _count(False)
    y = F.group_norm(y, norm.num_groups, norm.weight, norm.bias, norm.eps)
    y = _mish(y) * m
    return y if tb is None else y + tb[:, :, None, None]


def time_terms(blocks, temb):
    """ResnetBlock.mlp (Mish -> Linear, diffusion.py:66-67,75) of every block in one pass: Mish(temb) once, one Linear over the
    concatenated weights; returns the per-block [B, dim_out] terms (views).  The 12 blocks share the same input, and 36 separate
    [16 x 64] GEMMs and 80 elementwise launches cost more than the attention layers."""
    mt = _mish(temb)
    lins = [rb.mlp[1] for rb in blocks]
    tb = F.linear(mt, torch.cat([l.weight for l in lins], dim=0), torch.cat([l.bias for l in lins], dim=0))
    return list(torch.split(tb, [l.out_features for l in lins], dim=1))


def resnet(rb, v, m, temb, v1=None, tb=None):
    """ResnetBlock.forward (diffusion.py:73-78) on v, or on cat(v, v1) without materialising it.  tb: the block's time term if the
    caller has computed it (time_terms)."""
    if tb is None:
        lin = rb.mlp[1]
        tb = F.linear(_mish(temb), lin.weight, lin.bias)
    h = _conv_gn_mish(rb.block1, v, m, tb=tb, v1=v1)
    h = _conv_gn_mish(rb.block2, h, m)
    if isinstance(rb.res_conv, torch.nn.Conv2d):
        rc = rb.res_conv
        if v1 is None:
            r = _conv1x1(v, m, rc)
        elif _hip(v) and backend().conv1x1_supported(v.shape[1], rc.out_channels) and backend().conv1x1_supported(v1.shape[1], rc.out_channels):
            # a 1x1 convolution of a concatenation is the sum of the convolutions of its parts with the weight's column blocks
            c0 = v.shape[1]
            _count(True)
            r = MaskedConv1x1.apply(v.contiguous(), m, rc.weight[:, :c0].contiguous(), rc.bias)
            r = MaskedResidualAdd.apply(r, MaskedConv1x1.apply(v1.contiguous(), m, rc.weight[:, c0:].contiguous(), None), None)
        else:
            _count(False)
            r = F.conv2d(torch.cat((v, v1), dim=1) * m, rc.weight, rc.bias)
        return MaskedResidualAdd.apply(h, r.contiguous(), None) if _hip(h) else h + r
    if v1 is not None:
        v = torch.cat((v, v1), dim=1)
    return MaskedResidualAdd.apply(h, v.contiguous(), m) if _hip(h) else h + v * m


def _resample(v, m, conv, up):
    if _hip(v) and backend().resample_supported(v.shape[1], conv.out_channels, v.shape[2], v.shape[3], up, B=v.shape[0]):
        return ResampleConv.apply(v.contiguous(), m, conv.weight, conv.bias, up)
    return conv(v * m)


def attention(res, v):
    rez = res.fn
    att = rez.fn
    b, c, hh, ww = v.shape
    qkv = _conv1x1(v, None, att.to_qkv)
    hip = (not FORCE_TORCH and v.is_cuda and v.dtype == torch.float32 and att.heads == 4 and qkv.shape[1] == 384 and
           v.numel() % 4 == 0)
    if hip:
        out = LinearAttentionCore.apply(qkv.contiguous())
    else:
        q, k, val = qkv.view(b, 3, att.heads, -1, hh * ww).unbind(1)
        k = torch.softmax(k, dim=-1)
        ctx = torch.matmul(k, val.transpose(-1, -2))            # [b, heads, d, e]
        out = torch.matmul(ctx.transpose(-1, -2), q)            # [b, heads, e, n]
        out = out.reshape(b, -1, hh, ww)
    y = _conv1x1(out, None, att.to_out)
    if hip:
        return RezeroResidual.apply(y.contiguous(), rez.g, v.contiguous())
    return y * rez.g + v


def time_embedding(est, t):
    half = est.dim // 2
    freq = torch.exp(torch.arange(half, device=t.device).float() * -(math.log(10000) / (half - 1)))
    arg = est.pe_scale * t[:, None] * freq[None, :]
    emb = torch.cat((arg.sin(), arg.cos()), dim=-1)
    l0, l2 = est.mlp[0], est.mlp[2]
    return F.linear(_mish(F.linear(emb, l0.weight, l0.bias)), l2.weight, l2.bias)


def _pack_specs(est):
    """Every convolution weight the HIP training kernels of one step multiply with, in the forms they need (forward packing and,
    where a data gradient is taken, the transposed one): the argument of backend().prepack.  First layers (2 / 3 stacked input
    planes) and Upsample's re-indexed gradient weight pack themselves where they are used."""
    be = backend()
    cached = est.__dict__.get("_gtts_pack_specs")
    if cached is not None and all(m.weight is s[0] for m, s in zip(cached.mods, cached)):
        return cached                                     # (the module tree is static; a Parameter that was REPLACED rebuilds the list)
    specs = be.PackSpecs()
    specs.mods = []
    for mod in est.modules():
        n0 = len(specs)
        if isinstance(mod, torch.nn.ConvTranspose2d):
            ci, co = mod.in_channels, mod.out_channels
            if mod.kernel_size == (4, 4) and be.resample_supported(ci, co, 2, 2, True):
                specs.append((mod.weight, ci, co, False, "up"))
        elif isinstance(mod, torch.nn.Conv2d):
            ci, co = mod.in_channels, mod.out_channels
            if mod.kernel_size == (3, 3) and mod.stride == (2, 2):
                if be.resample_supported(ci, co, 2, 2, False):
                    specs.append((mod.weight, ci, co, False, "dn"))
                    specs.append((mod.weight, co, ci, False, "dn_T"))
            elif mod.kernel_size == (3, 3) and mod.stride == (1, 1) and ci >= 16:
                if be.conv3x3_supported(ci, co, need_dgrad=True):
                    specs.append((mod.weight, ci, co, False, "3x3"))
                    specs.append((mod.weight, co, ci, True, "3x3"))
            elif mod.kernel_size == (1, 1) and ci >= 16:
                if be.conv1x1_supported(ci, co, need_dgrad=True):
                    specs.append((mod.weight, ci, co, False, "1x1"))
                    specs.append((mod.weight, co, ci, True, "1x1"))
        specs.mods.extend([mod] * (len(specs) - n0))
    est.__dict__["_gtts_pack_specs"] = specs
    return specs


def estimator(est, x, mask, mu, t, spk=None):
    if not FORCE_TORCH and x.is_cuda:
        be = backend()
        be.new_pack_generation()               # packed weight copies live for this call's forward + backward only ...
        be.prepack(_pack_specs(est))           # ... and all of them are made here, in one launch (also under no_grad: a validation
                                               # call would otherwise pack ~45 weights one by one, each with its own allocation)
    temb = time_embedding(est, t)
    planes = [mu, x]
    if est.n_spks >= 2:
        l0, l2 = est.spk_mlp[0], est.spk_mlp[2]
        s = F.linear(_mish(F.linear(spk, l0.weight, l0.bias)), l2.weight, l2.bias)
        planes.append(s[:, :, None].expand(-1, -1, x.shape[-1]))
    v = torch.stack(planes, 1)
    m = mask[:, None]
    blocks = [rb for r1, r2, _, _ in est.downs for rb in (r1, r2)] + [est.mid_block1, est.mid_block2] + \
             [rb for r1, r2, _, _ in est.ups for rb in (r1, r2)]
    tbs = iter(time_terms(blocks, temb))
    skips, pyramid = [], [m]
    for r1, r2, att, down in est.downs:
        m = pyramid[-1]
        v = resnet(r1, v, m, temb, tb=next(tbs))
        v = attention(att, resnet(r2, v, m, temb, tb=next(tbs)))
        skips.append(v)
        if not isinstance(down, torch.nn.Identity):
            v = _resample(v, m, down.conv, False)
        pyramid.append(m[..., ::2].contiguous())        # (contiguous: every op below takes its [B, W] view without a copy)
    pyramid.pop()
    m = pyramid[-1]
    v = resnet(est.mid_block1, v, m, temb, tb=next(tbs))
    v = resnet(est.mid_block2, attention(est.mid_attn, v), m, temb, tb=next(tbs))
    for r1, r2, att, up in est.ups:
        m = pyramid.pop()
        v = resnet(r1, v, m, temb, v1=skips.pop(), tb=next(tbs))       # (torch.cat read in place)
        v = attention(att, resnet(r2, v, m, temb, tb=next(tbs)))
        v = _resample(v, m, up.conv, True)
    m = mask[:, None]
    v = _conv_gn_mish(est.final_block, v, m)
    return final_conv(est.final_conv, v, m)


def final_conv(fc, v, m):
    """final_conv(v * mask) * mask, squeezed (diffusion.py:214-216; DiffVC/model/diffusion.py:104-106)."""
    if _hip(v) and fc.out_channels == 1:
        _count(True)
        return FinalConv.apply(v.contiguous(), m, fc.weight, fc.bias).squeeze(1)
    _count(False)
    out = F.conv2d(v * m, fc.weight, fc.bias)
    return (out * m).squeeze(1)
