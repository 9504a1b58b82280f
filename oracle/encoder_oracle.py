"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- functional CPU restatement of the encoders either side of the
sampling path: Grad-TTS TextEncoder (Grad-TTS/model/text_encoder.py:281-326) and DiffVC MelEncoder
(DiffVC/model/encoder.py:257-284), inference mode (dropout = identity), over a plain state_dict.

Stock torch CPU fp32 ops are the reference's arithmetic.  The relative-position attention follows the reference's own
pad-and-reshape re-indexing (text_encoder.py:177-199).  Pinned by tests/golden/encoder.npz (outputs of the reference's
modules, tests/golden/make_golden_encoder.py) and live in tests/test_encoder_cpu.py.
"""
import math

import torch
import torch.nn.functional as F


def sequence_mask(length, max_length=None):
    """Grad-TTS/model/utils.py:6-10."""
    if max_length is None:
        max_length = length.max()
    return torch.arange(int(max_length), dtype=length.dtype).unsqueeze(0) < length.unsqueeze(1)


def layer_norm(sd, p, x, eps=1e-4):
    """LayerNorm.forward, text_encoder.py:20-27 (over dim 1, biased variance)."""
    mean = torch.mean(x, 1, keepdim=True)
    var = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + eps)
    return x * sd[p + "gamma"].view(1, -1, 1) + sd[p + "beta"].view(1, -1, 1)


def conv(sd, p, x, k):
    return F.conv1d(x, sd[p + "weight"], sd[p + "bias"], padding=k // 2)


def prenet(sd, p, x, x_mask):
    """ConvReluNorm.forward, text_encoder.py:54-61 (kernel 5, 3 layers)."""
    x_org = x
    for i in range(3):
        x = conv(sd, p + "conv_layers.%d." % i, x * x_mask, 5)
        x = layer_norm(sd, p + "norm_layers.%d." % i, x)
        x = torch.relu(x)
    x = x_org + conv(sd, p + "proj.", x, 1)
    return x * x_mask


def _rel_embeddings(emb, length, window):
    """_get_relative_embeddings, text_encoder.py:185-199: zero-pad the 2w+1 embeddings to 2*length-1 relative positions."""
    pad = max(length - (window + 1), 0)
    start = max((window + 1) - length, 0)
    if pad > 0:
        emb = F.pad(emb, [0, 0, pad, pad, 0, 0])
    return emb[:, start:start + 2 * length - 1]


def _rel_to_abs(x):
    """_relative_position_to_absolute_position, text_encoder.py:201-207: [b,h,l,2l-1] -> [b,h,l,l]."""
    b, h, l, _ = x.shape
    x = F.pad(x, [0, 1, 0, 0, 0, 0, 0, 0])
    flat = F.pad(x.view(b, h, l * 2 * l), [0, l - 1, 0, 0, 0, 0])
    return flat.view(b, h, l + 1, 2 * l - 1)[:, :, :l, l - 1:]


def _abs_to_rel(x):
    """_absolute_position_to_relative_position, text_encoder.py:209-215: [b,h,l,l] -> [b,h,l,2l-1]."""
    b, h, l, _ = x.shape
    x = F.pad(x, [0, l - 1, 0, 0, 0, 0, 0, 0])
    flat = F.pad(x.view(b, h, l * l + l * (l - 1)), [l, 0, 0, 0, 0, 0])
    return flat.view(b, h, l, 2 * l)[:, :, :, 1:]


def attention(sd, p, x, attn_mask, n_heads, window):
    """MultiHeadAttention.forward / attention, text_encoder.py:136-175 (self-attention, relative window)."""
    q, k, v = conv(sd, p + "conv_q.", x, 1), conv(sd, p + "conv_k.", x, 1), conv(sd, p + "conv_v.", x, 1)
    b, d, t = k.shape
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t).transpose(2, 3)
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if window:
        ek = _rel_embeddings(sd[p + "emb_rel_k"], t, window)
        rel = torch.matmul(q, ek.unsqueeze(0).transpose(-2, -1))
        scores = scores + _rel_to_abs(rel) / math.sqrt(dk)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    pa = F.softmax(scores, dim=-1)
    out = torch.matmul(pa, v)
    if window:
        ev = _rel_embeddings(sd[p + "emb_rel_v"], t, window)
        out = out + torch.matmul(_abs_to_rel(pa), ev.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return conv(sd, p + "conv_o.", out, 1)


def ffn(sd, p, x, x_mask, k):
    """FFN.forward, text_encoder.py:230-237."""
    x = conv(sd, p + "conv_1.", x * x_mask, k)
    x = torch.relu(x)
    x = conv(sd, p + "conv_2.", x * x_mask, k)
    return x * x_mask


def encoder(sd, p, x, x_mask, n_layers, n_heads, window, k):
    """Encoder.forward, text_encoder.py:264-278."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    for i in range(n_layers):
        x = x * x_mask
        y = attention(sd, p + "attn_layers.%d." % i, x, attn_mask, n_heads, window)
        x = layer_norm(sd, p + "norm_layers_1.%d." % i, x + y)
        y = ffn(sd, p + "ffn_layers.%d." % i, x, x_mask, k)
        x = layer_norm(sd, p + "norm_layers_2.%d." % i, x + y)
    return x * x_mask


def duration_predictor(sd, p, x, x_mask, k):
    """DurationPredictor.forward, text_encoder.py:84-97."""
    x = conv(sd, p + "conv_1.", x * x_mask, k)
    x = layer_norm(sd, p + "norm_1.", torch.relu(x))
    x = conv(sd, p + "conv_2.", x * x_mask, k)
    x = layer_norm(sd, p + "norm_2.", torch.relu(x))
    x = conv(sd, p + "proj.", x * x_mask, 1)
    return x * x_mask


def text_encoder_forward(sd, ids, lengths, n_heads=2, window=4, k=3):
    """TextEncoder.forward, text_encoder.py:310-326 (n_spks = 1, as GradTTS builds it)."""
    C = sd["emb.weight"].shape[1]
    n_layers = sum(1 for key in sd if key.startswith("encoder.norm_layers_1.") and key.endswith("gamma"))
    x = F.embedding(ids, sd["emb.weight"]) * math.sqrt(C)
    x = torch.transpose(x, 1, -1)
    x_mask = sequence_mask(lengths, x.size(2)).unsqueeze(1).to(x.dtype)
    x = prenet(sd, "prenet.", x, x_mask)
    x = encoder(sd, "encoder.", x, x_mask, n_layers, n_heads, window, k)
    mu = conv(sd, "proj_m.", x, 1) * x_mask
    logw = duration_predictor(sd, "proj_w.", x, x_mask, k)
    return mu, logw, x_mask


def mel_encoder_forward(sd, x, x_mask, n_heads=2, window=4, k=3):
    """MelEncoder.forward, DiffVC/model/encoder.py:279-284."""
    n_layers = sum(1 for key in sd if key.startswith("encoder.norm_layers_1.") and key.endswith("gamma"))
    x = conv(sd, "init_proj.", x * x_mask, 1)
    x = prenet(sd, "prenet.", x, x_mask)
    x = encoder(sd, "encoder.", x, x_mask, n_layers, n_heads, window, k)
    return conv(sd, "term_proj.", x * x_mask, 1)


def make_state(mode="text", n_vocab=149, n_feats=80, C=192, filt=768, filt_dp=256, n_heads=2, n_layers=6, k=3, window=4, seed=0):
    """Random weights in the reference's state_dict layout; every tensor non-trivial (the reference zero-initialises
    prenet.proj, which would hide the prenet)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    def cv(name, cout, cin, kk):
        sd[name + ".weight"] = uni((cout, cin, kk), cin * kk)
        sd[name + ".bias"] = uni((cout,), cin * kk)

    def ln(name, c):
        sd[name + ".gamma"] = 1.0 + 0.2 * (torch.rand(c, generator=g) - 0.5)
        sd[name + ".beta"] = 0.2 * (torch.rand(c, generator=g) - 0.5)

    if mode == "text":
        sd["emb.weight"] = torch.randn(n_vocab, C, generator=g) * C ** -0.5
    else:
        cv("init_proj", C, n_feats, 1)
    for i in range(3):
        cv("prenet.conv_layers.%d" % i, C, C, 5)
    for i in range(3):
        ln("prenet.norm_layers.%d" % i, C)
    cv("prenet.proj", C, C, 1)
    dk = C // n_heads
    for i in range(n_layers):
        p = "encoder.attn_layers.%d." % i
        sd[p + "emb_rel_k"] = torch.randn(1, 2 * window + 1, dk, generator=g) * dk ** -0.5
        sd[p + "emb_rel_v"] = torch.randn(1, 2 * window + 1, dk, generator=g) * dk ** -0.5
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            cv(p + n, C, C, 1)
    for i in range(n_layers):
        ln("encoder.norm_layers_1.%d" % i, C)
    for i in range(n_layers):
        cv("encoder.ffn_layers.%d.conv_1" % i, filt, C, k)
        cv("encoder.ffn_layers.%d.conv_2" % i, C, filt, k)
    for i in range(n_layers):
        ln("encoder.norm_layers_2.%d" % i, C)
    if mode == "text":
        cv("proj_m", n_feats, C, 1)
        cv("proj_w.conv_1", filt_dp, C, k)
        ln("proj_w.norm_1", filt_dp)
        cv("proj_w.conv_2", filt_dp, filt_dp, k)
        ln("proj_w.norm_2", filt_dp)
        cv("proj_w.proj", 1, filt_dp, 1)
    else:
        cv("term_proj", n_feats, C, 1)
    return sd
