"""Text encoder + duration predictor with the reference's parameter layout (Grad-TTS/model/text_encoder.py).

Inference (eval mode, torch.no_grad, HIP tensors) runs the kernels of csrc/enc.hip through gtts_enc_forward (SURVEY.md
section 8f rank 4); training composes the torch modules below over the same parameters.  Written from the behaviour of
the reference modules; parameter names / shapes are identical so `load_state_dict(strict=True)` of reference checkpoints
works:

  emb, prenet.{conv_layers,norm_layers}.{0,1,2}, prenet.proj, encoder.{attn_layers,norm_layers_1,ffn_layers,
  norm_layers_2}.{i}, proj_m, proj_w.{conv_1,norm_1,conv_2,norm_2,proj}

The windowed relative-position attention (text_encoder.py:143-210) is expressed here with band gathers instead of
the reference's pad-and-reshape trick; both pick, for query i and key j with |j - i| <= window, the learned
embedding number (j - i + window), and zero outside the window.
"""
import math

import torch
import torch.nn.functional as F

from .base import BaseModule
from .utils import sequence_mask


class LayerNorm(BaseModule):
    """Channel-wise layer norm over dim 1, eps 1e-4, parameters `gamma` / `beta` (text_encoder.py:11-29)."""

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = torch.nn.Parameter(torch.ones(channels))
        self.beta = torch.nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        mu = x.mean(1, keepdim=True)
        var = ((x - mu) ** 2).mean(1, keepdim=True)
        y = (x - mu) * torch.rsqrt(var + self.eps)
        bshape = [1, -1] + [1] * (x.dim() - 2)
        return y * self.gamma.view(bshape) + self.beta.view(bshape)


class ConvReluNorm(BaseModule):
    """Prenet: n_layers x (conv -> LayerNorm -> ReLU -> dropout), zero-initialised 1x1 proj, residual
    (text_encoder.py:32-64)."""

    def __init__(self, in_channels, hidden_channels, out_channels, kernel_size, n_layers, p_dropout):
        super().__init__()
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.kernel_size, self.n_layers, self.p_dropout = kernel_size, n_layers, p_dropout
        self.conv_layers = torch.nn.ModuleList()
        self.norm_layers = torch.nn.ModuleList()
        for i in range(n_layers):
            cin = in_channels if i == 0 else hidden_channels
            self.conv_layers.append(torch.nn.Conv1d(cin, hidden_channels, kernel_size, padding=kernel_size // 2))
            self.norm_layers.append(LayerNorm(hidden_channels))
        self.relu_drop = torch.nn.Sequential(torch.nn.ReLU(), torch.nn.Dropout(p_dropout))
        self.proj = torch.nn.Conv1d(hidden_channels, out_channels, 1)
        torch.nn.init.zeros_(self.proj.weight)
        torch.nn.init.zeros_(self.proj.bias)

    def forward(self, x, x_mask):
        y = x
        for conv, norm in zip(self.conv_layers, self.norm_layers):
            y = self.relu_drop(norm(conv(y * x_mask)))
        return (x + self.proj(y)) * x_mask


class DurationPredictor(BaseModule):
    """text_encoder.py:67-93."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout):
        super().__init__()
        self.in_channels, self.filter_channels, self.p_dropout = in_channels, filter_channels, p_dropout
        self.drop = torch.nn.Dropout(p_dropout)
        self.conv_1 = torch.nn.Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_1 = LayerNorm(filter_channels)
        self.conv_2 = torch.nn.Conv1d(filter_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.norm_2 = LayerNorm(filter_channels)
        self.proj = torch.nn.Conv1d(filter_channels, 1, 1)

    def forward(self, x, x_mask):
        x = self.drop(self.norm_1(torch.relu(self.conv_1(x * x_mask))))
        x = self.drop(self.norm_2(torch.relu(self.conv_2(x * x_mask))))
        return self.proj(x * x_mask) * x_mask


class MultiHeadAttention(BaseModule):
    """Self-attention with learned relative-position keys/values inside a +-window band
    (text_encoder.py:96-215; heads share the relative embeddings)."""

    def __init__(self, channels, out_channels, n_heads, window_size=None, heads_share=True, p_dropout=0.0,
                 proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.window_size, self.heads_share, self.proximal_bias, self.p_dropout = (window_size, heads_share,
                                                                                proximal_bias, p_dropout)
        self.attn = None
        self.k_channels = channels // n_heads
        self.conv_q = torch.nn.Conv1d(channels, channels, 1)
        self.conv_k = torch.nn.Conv1d(channels, channels, 1)
        self.conv_v = torch.nn.Conv1d(channels, channels, 1)
        if window_size is not None:
            n_rel = 1 if heads_share else n_heads
            std = self.k_channels ** -0.5
            self.emb_rel_k = torch.nn.Parameter(torch.randn(n_rel, 2 * window_size + 1, self.k_channels) * std)
            self.emb_rel_v = torch.nn.Parameter(torch.randn(n_rel, 2 * window_size + 1, self.k_channels) * std)
        self.conv_o = torch.nn.Conv1d(channels, out_channels, 1)
        self.drop = torch.nn.Dropout(p_dropout)
        torch.nn.init.xavier_uniform_(self.conv_q.weight)
        torch.nn.init.xavier_uniform_(self.conv_k.weight)
        if proximal_init:
            self.conv_k.weight.data.copy_(self.conv_q.weight.data)
            self.conv_k.bias.data.copy_(self.conv_q.bias.data)
        torch.nn.init.xavier_uniform_(self.conv_v.weight)

    def _band(self, t, device):
        """rel[i, j] = j - i + window (clamped) and whether |j - i| <= window."""
        w = self.window_size
        pos = torch.arange(t, device=device)
        rel = pos[None, :] - pos[:, None] + w
        return rel.clamp(0, 2 * w), (rel >= 0) & (rel <= 2 * w)

    def attention(self, query, key, value, mask=None):
        b, d, t_s = key.shape
        t_t = query.shape[2]
        h, dk = self.n_heads, self.k_channels
        q = query.view(b, h, dk, t_t).transpose(2, 3)
        k = key.view(b, h, dk, t_s).transpose(2, 3)
        v = value.view(b, h, dk, t_s).transpose(2, 3)
        scale = math.sqrt(dk)
        scores = torch.matmul(q, k.transpose(-2, -1)) / scale
        if self.window_size is not None:
            assert t_s == t_t, "Relative attention is only available for self-attention."
            rel, inside = self._band(t_s, q.device)
            # logits against every relative embedding, then pick column (j - i + w) for each (i, j) in the band
            rel_logits = torch.matmul(q, self.emb_rel_k.unsqueeze(0).transpose(-2, -1))      # [b,h,t,2w+1]
            local = rel_logits.gather(-1, rel.expand(b, h, t_s, t_s)) * inside.to(q.dtype)
            scores = scores + local / scale
        if self.proximal_bias:
            assert t_s == t_t, "Proximal bias is only available for self-attention."
            r = torch.arange(t_s, dtype=torch.float32, device=scores.device)
            scores = scores - torch.log1p((r[None, :] - r[:, None]).abs())[None, None]
        if mask is not None:
            scores = scores.masked_fill(mask == 0, -1e4)
        p_attn = self.drop(F.softmax(scores, dim=-1))
        out = torch.matmul(p_attn, v)
        if self.window_size is not None:
            w = self.window_size
            # weights of the 2w+1 relative offsets: relw[i, r] = p[i, i + r - w] (zero outside the sequence)
            pos = torch.arange(t_s, device=q.device)
            col = pos[:, None] + torch.arange(2 * w + 1, device=q.device)[None, :] - w
            ok = (col >= 0) & (col < t_s)
            relw = p_attn.gather(-1, col.clamp(0, t_s - 1).expand(b, h, t_s, 2 * w + 1)) * ok.to(q.dtype)
            out = out + torch.matmul(relw, self.emb_rel_v.unsqueeze(0))
        out = out.transpose(2, 3).contiguous().view(b, d, t_t)
        return out, p_attn

    def forward(self, x, c, attn_mask=None):
        y, self.attn = self.attention(self.conv_q(x), self.conv_k(c), self.conv_v(c), mask=attn_mask)
        return self.conv_o(y)


class FFN(BaseModule):
    """text_encoder.py:218-239."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0):
        super().__init__()
        self.in_channels, self.out_channels, self.filter_channels = in_channels, out_channels, filter_channels
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        self.conv_1 = torch.nn.Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.conv_2 = torch.nn.Conv1d(filter_channels, out_channels, kernel_size, padding=kernel_size // 2)
        self.drop = torch.nn.Dropout(p_dropout)

    def forward(self, x, x_mask):
        y = self.drop(torch.relu(self.conv_1(x * x_mask)))
        return self.conv_2(y * x_mask) * x_mask


class Encoder(BaseModule):
    """n_layers x (rel-pos MHA + FFN), post-norm (text_encoder.py:242-278)."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0,
                 window_size=None, **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels, self.n_heads = hidden_channels, filter_channels, n_heads
        self.n_layers, self.kernel_size, self.p_dropout, self.window_size = n_layers, kernel_size, p_dropout, window_size
        self.drop = torch.nn.Dropout(p_dropout)
        self.attn_layers = torch.nn.ModuleList()
        self.norm_layers_1 = torch.nn.ModuleList()
        self.ffn_layers = torch.nn.ModuleList()
        self.norm_layers_2 = torch.nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads,
                                                       window_size=window_size, p_dropout=p_dropout))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                       p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, x_mask):
        pair_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        for attn, n1, ffn, n2 in zip(self.attn_layers, self.norm_layers_1, self.ffn_layers, self.norm_layers_2):
            x = x * x_mask
            x = n1(x + self.drop(attn(x, x, pair_mask)))
            x = n2(x + self.drop(ffn(x, x_mask)))
        return x * x_mask


class TextEncoder(BaseModule):
    """text_encoder.py:281-326.  Note the reference quirk: GradTTS builds it with the default n_spks=1, so the
    speaker embedding never reaches the encoder (tts.py:45-47)."""

    def __init__(self, n_vocab, n_feats, n_channels, filter_channels, filter_channels_dp, n_heads, n_layers,
                 kernel_size, p_dropout, window_size=None, spk_emb_dim=64, n_spks=1):
        super().__init__()
        self.n_vocab, self.n_feats, self.n_channels = n_vocab, n_feats, n_channels
        self.filter_channels, self.filter_channels_dp = filter_channels, filter_channels_dp
        self.n_heads, self.n_layers, self.kernel_size = n_heads, n_layers, kernel_size
        self.p_dropout, self.window_size, self.spk_emb_dim, self.n_spks = p_dropout, window_size, spk_emb_dim, n_spks

        self.emb = torch.nn.Embedding(n_vocab, n_channels)
        torch.nn.init.normal_(self.emb.weight, 0.0, n_channels ** -0.5)
        self.prenet = ConvReluNorm(n_channels, n_channels, n_channels, kernel_size=5, n_layers=3, p_dropout=0.5)
        width = n_channels + (spk_emb_dim if n_spks > 1 else 0)
        self.encoder = Encoder(width, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                               window_size=window_size)
        self.proj_m = torch.nn.Conv1d(width, n_feats, 1)
        self.proj_w = DurationPredictor(width, filter_channels_dp, kernel_size, p_dropout)
        self._hip_enc = None
        self._hip_blob = None
        self._hip_key = None

    def invalidate_packed(self):
        self._hip_blob = None
        self._hip_key = None

    def _hip_forward(self, x, x_lengths):
        from ._backend import backend
        be = backend()
        if self._hip_enc is None:
            self._hip_enc = be.Encoder("text", self.n_vocab, self.n_feats, self.n_channels, self.filter_channels,
                                       self.filter_channels_dp, self.n_heads, self.n_layers, self.kernel_size, self.window_size)
        params = list(self.named_parameters())
        key = (str(x.device),) + tuple((p.data_ptr(), p._version) for _, p in params)
        if self._hip_blob is None or self._hip_key != key:
            self._hip_blob = self._hip_enc.pack({n: p for n, p in params}, x.device)
            self._hip_key = key
        x_mask = sequence_mask(x_lengths, x.shape[1]).unsqueeze(1).to(torch.float32)
        mu, logw = self._hip_enc.forward(self._hip_blob, x, x_mask)
        return mu, logw, x_mask

    def forward(self, x, x_lengths, spk=None):
        if x.is_cuda and not torch.is_grad_enabled() and not self.training and self.n_spks == 1:
            return self._hip_forward(x, x_lengths)
        h = (self.emb(x) * math.sqrt(self.n_channels)).transpose(1, -1)
        x_mask = sequence_mask(x_lengths, h.size(2)).unsqueeze(1).to(h.dtype)
        h = self.prenet(h, x_mask)
        if self.n_spks > 1:
            h = torch.cat([h, spk.unsqueeze(-1).repeat(1, 1, h.shape[-1])], dim=1)
        h = self.encoder(h, x_mask)
        mu = self.proj_m(h) * x_mask
        logw = self.proj_w(h.detach(), x_mask)
        return mu, logw, x_mask
