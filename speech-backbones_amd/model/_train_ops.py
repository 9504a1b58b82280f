"""Autograd composition of the score U-Net out of stock torch ops -- TRAINING ONLY.

The sampling path (torch.no_grad) never comes here: it runs the hand-written HIP kernels behind the C ABI and
raises if they are unavailable.  Training (Diffusion.compute_loss -> loss_t -> estimator with autograd,
Grad-TTS/model/diffusion.py:281-294) needs gradients w.r.t. the same nn.Parameters; until the backward
kernels exist it is expressed with PyTorch-ROCm's differentiable ops here.
"""
import math

import torch
import torch.nn.functional as F


def _mish(v):
    return v * torch.tanh(F.softplus(v))


def _conv_gn_mish(blk, v, m):
    conv, norm = blk.block[0], blk.block[1]
    y = F.conv2d(v * m, conv.weight, conv.bias, padding=1)
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
