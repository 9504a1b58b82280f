"""Autograd composition of the score U-Net -- TRAINING ONLY (the sampling path under torch.no_grad never comes here).

Training (Diffusion.compute_loss -> loss_t -> estimator with autograd, Grad-TTS/model/diffusion.py:281-294) needs gradients
w.r.t. the same nn.Parameters.  On HIP tensors the 3x3 Block convolutions -- 84 % of the network's FLOPs, forward and
backward -- run on the hand-written kernels of csrc/train.hip (forward and data gradient on the inference MFMA kernel,
weight gradient as an MFMA reduction over pixels) through `MaskedConv3x3`, every Block's GroupNorm + Mish + mask (forward and
backward fused, csrc/train_norm.hip) through `GnMishMask`, and the loss head through `ScoreLoss`; attention, the 1x1 / resampling
convolutions and the small MLPs are PyTorch-ROCm differentiable ops.
On CPU tensors (tests) everything is stock torch.
"""
import math

import torch
import torch.nn.functional as F

from ._backend import backend


class MaskedConv3x3(torch.autograd.Function):
    """y = Conv2d_3x3(x * mask) + bias with all three gradients on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, mask, weight, bias):
        be = backend()
        cols = mask.reshape(mask.shape[0], mask.shape[-1])          # [B,1,1,W] -> [B,W]
        ctx.save_for_backward(x, cols, weight)
        return be.conv3x3_masked(x, cols, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        be = backend()
        x, cols, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = be.conv3x3_dgrad(dy, weight) * cols[:, None, None, :]
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:      # (one kernel produces both)
            dw, db = be.conv3x3_wgrad(x, cols, dy)
        return dx, None, (dw if ctx.needs_input_grad[2] else None), (db if ctx.needs_input_grad[3] else None)


class GnMishMask(torch.autograd.Function):
    """Mish(GroupNorm(y)) * mask (Block.forward, diffusion.py:53-58) with forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, y, mask, gamma, beta, groups, eps):
        be = backend()
        cols = mask.reshape(mask.shape[0], mask.shape[-1])          # [B,1,1,W] -> [B,W]
        out, stats = be.gn_mish_forward(y, gamma, beta, cols, groups, eps)
        ctx.save_for_backward(y, cols, gamma, beta, stats)
        ctx.groups = groups
        return out

    @staticmethod
    def backward(ctx, dout):
        y, cols, gamma, beta, stats = ctx.saved_tensors
        dy, dg, db = backend().gn_mish_backward(dout.contiguous(), y, gamma, beta, cols, stats, ctx.groups)
        return dy, None, dg, db, None, None


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


def _hip_conv_ok(v, conv):
    return (not FORCE_TORCH and v.is_cuda and v.dtype == torch.float32 and conv.kernel_size == (3, 3) and
            backend().conv3x3_supported(conv.in_channels, conv.out_channels))


def _mish(v):
    return v * torch.tanh(F.softplus(v))


def _conv_gn_mish(blk, v, m):
    conv, norm = blk.block[0], blk.block[1]
    if _hip_conv_ok(v, conv):
        y = MaskedConv3x3.apply(v.contiguous(), m, conv.weight, conv.bias)
    else:
        y = F.conv2d(v * m, conv.weight, conv.bias, padding=1)
    if y.is_cuda and y.dtype == torch.float32 and not FORCE_TORCH and y.dim() == 4 and y.shape[1] % norm.num_groups == 0:
        return GnMishMask.apply(y.contiguous(), m, norm.weight, norm.bias, norm.num_groups, norm.eps)
    y = F.group_norm(y, norm.num_groups, norm.weight, norm.bias, norm.eps)
    return _mish(y) * m


def resnet(rb, v, m, temb):
    lin = rb.mlp[1]
    h = _conv_gn_mish(rb.block1, v, m)
    h = h + F.linear(_mish(temb), lin.weight, lin.bias)[:, :, None, None]
    h = _conv_gn_mish(rb.block2, h, m)
    if isinstance(rb.res_conv, torch.nn.Conv2d):
        return h + F.conv2d(v * m, rb.res_conv.weight, rb.res_conv.bias)
    return h + v * m


def attention(res, v):
    rez = res.fn
    att = rez.fn
    b, c, hh, ww = v.shape
    qkv = F.conv2d(v, att.to_qkv.weight).view(b, 3, att.heads, -1, hh * ww)
    q, k, val = qkv.unbind(1)
    k = torch.softmax(k, dim=-1)
    ctx = torch.matmul(k, val.transpose(-1, -2))            # [b, heads, d, e]
    out = torch.matmul(ctx.transpose(-1, -2), q)            # [b, heads, e, n]
    out = out.reshape(b, -1, hh, ww)
    y = F.conv2d(out, att.to_out.weight, att.to_out.bias)
    return y * rez.g + v


def time_embedding(est, t):
    half = est.dim // 2
    freq = torch.exp(torch.arange(half, device=t.device).float() * -(math.log(10000) / (half - 1)))
    arg = est.pe_scale * t[:, None] * freq[None, :]
    emb = torch.cat((arg.sin(), arg.cos()), dim=-1)
    l0, l2 = est.mlp[0], est.mlp[2]
    return F.linear(_mish(F.linear(emb, l0.weight, l0.bias)), l2.weight, l2.bias)


def estimator(est, x, mask, mu, t, spk=None):
    temb = time_embedding(est, t)
    planes = [mu, x]
    if est.n_spks >= 2:
        l0, l2 = est.spk_mlp[0], est.spk_mlp[2]
        s = F.linear(_mish(F.linear(spk, l0.weight, l0.bias)), l2.weight, l2.bias)
        planes.append(s[:, :, None].expand(-1, -1, x.shape[-1]))
    v = torch.stack(planes, 1)
    m = mask[:, None]
    skips, pyramid = [], [m]
    for r1, r2, att, down in est.downs:
        m = pyramid[-1]
        v = attention(att, resnet(r2, resnet(r1, v, m, temb), m, temb))
        skips.append(v)
        if not isinstance(down, torch.nn.Identity):
            v = down.conv(v * m)
        pyramid.append(m[..., ::2])
    pyramid.pop()
    m = pyramid[-1]
    v = resnet(est.mid_block2, attention(est.mid_attn, resnet(est.mid_block1, v, m, temb)), m, temb)
    for r1, r2, att, up in est.ups:
        m = pyramid.pop()
        v = torch.cat((v, skips.pop()), dim=1)
        v = attention(att, resnet(r2, resnet(r1, v, m, temb), m, temb))
        v = up.conv(v * m)
    m = mask[:, None]
    v = _conv_gn_mish(est.final_block, v, m)
    out = F.conv2d(v * m, est.final_conv.weight, est.final_conv.bias)
    return (out * m).squeeze(1)
