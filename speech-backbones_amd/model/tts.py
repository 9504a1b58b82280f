"""GradTTS -- drop-in for Grad-TTS/model/tts.py:21-181 (same constructor, forward and compute_loss signatures,
same state_dict).  The decoder (Diffusion) samples on the MI355X HIP kernels; MAS (training) runs on the GPU too.
"""
import math
import random

import torch

from . import monotonic_align
from ._backend import backend
from .base import BaseModule
from .diffusion import Diffusion
from .text_encoder import TextEncoder
from .utils import duration_loss, fix_len_compatibility, generate_path, sequence_mask


class GradTTS(BaseModule):
    def __init__(self, n_vocab, n_spks, spk_emb_dim, n_enc_channels, filter_channels, filter_channels_dp,
                 n_heads, n_enc_layers, enc_kernel, enc_dropout, window_size,
                 n_feats, dec_dim, beta_min, beta_max, pe_scale):
        super().__init__()
        for k, v in dict(n_vocab=n_vocab, n_spks=n_spks, spk_emb_dim=spk_emb_dim, n_enc_channels=n_enc_channels,
                         filter_channels=filter_channels, filter_channels_dp=filter_channels_dp, n_heads=n_heads,
                         n_enc_layers=n_enc_layers, enc_kernel=enc_kernel, enc_dropout=enc_dropout,
                         window_size=window_size, n_feats=n_feats, dec_dim=dec_dim, beta_min=beta_min,
                         beta_max=beta_max, pe_scale=pe_scale).items():
            setattr(self, k, v)
        if n_spks > 1:
            self.spk_emb = torch.nn.Embedding(n_spks, spk_emb_dim)
        # tts.py:45-47: the encoder is built with its default n_spks=1 (speaker embedding is ignored there)
        self.encoder = TextEncoder(n_vocab, n_feats, n_enc_channels, filter_channels, filter_channels_dp, n_heads,
                                   n_enc_layers, enc_kernel, enc_dropout, window_size)
        self.decoder = Diffusion(n_feats, dec_dim, n_spks, spk_emb_dim, beta_min, beta_max, pe_scale)

    @torch.no_grad()
    def forward(self, x, x_lengths, n_timesteps, temperature=1.0, stoc=False, spk=None, length_scale=1.0):
        """Text ids -> (encoder_outputs, decoder_outputs, attn)   (tts.py:50-99).

        x [B,t_x] int64 phoneme ids, x_lengths [B]; n_timesteps reverse-diffusion steps; temperature scales the
        terminal noise; stoc selects the SDE sampler; length_scale stretches durations."""
        x, x_lengths = self.relocate_input([x, x_lengths])
        if self.n_spks > 1:
            spk = self.spk_emb(spk)
        mu_x, logw, x_mask = self.encoder(x, x_lengths, spk)

        # durations -> frame counts (tts.py:77-81); one host sync for the padded length
        w_ceil = torch.ceil(torch.exp(logw) * x_mask) * length_scale
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        y_max_length = int(y_lengths.max())
        y_max_length_ = fix_len_compatibility(y_max_length)

        # alignment path from durations (tts.py:84-86), aligned prior mean (tts.py:89-91) and terminal sample (tts.py:94)
        y_mask = sequence_mask(y_lengths, y_max_length_).unsqueeze(1).to(x_mask.dtype)
        if mu_x.is_cuda and not torch.is_grad_enabled():
            # one HIP launch instead of the dense [B,t_x,T] path and its matmul.  The reference draws
            # randn_like(mu_y) on a transposed [B,T,80] buffer; the same strides keep the Philox stream -> element map.
            tmpl = torch.empty(mu_x.shape[0], y_max_length_, mu_x.shape[1], dtype=mu_x.dtype, device=mu_x.device)
            noise = torch.randn_like(tmpl.transpose(1, 2), device=mu_x.device)
            attn, mu_y, z = backend().expand_alignment(w_ceil.squeeze(1), x_mask.squeeze(1), y_lengths, mu_x, y_max_length_,
                                                      noise, temperature)
            attn = attn.unsqueeze(1)
        else:
            attn_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)
            attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
            mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
            z = mu_y + torch.randn_like(mu_y, device=mu_y.device) / temperature
        encoder_outputs = mu_y[:, :, :y_max_length]

        decoder_outputs = self.decoder(z, y_mask, mu_y, n_timesteps, stoc, spk)[:, :, :y_max_length]
        # reference quirk kept (tts.py:99): the slice below indexes the t_x axis of attn [B,1,t_x,T]
        return encoder_outputs, decoder_outputs, attn[:, :, :y_max_length]

    def compute_loss(self, x, x_lengths, y, y_lengths, spk=None, out_size=None):
        """(duration loss, prior loss, diffusion loss)   (tts.py:101-181)."""
        x, x_lengths, y, y_lengths = self.relocate_input([x, x_lengths, y, y_lengths])
        if self.n_spks > 1:
            spk = self.spk_emb(spk)
        mu_x, logw, x_mask = self.encoder(x, x_lengths, spk)
        y_max_length = y.shape[-1]
        y_mask = sequence_mask(y_lengths, y_max_length).unsqueeze(1).to(x_mask)
        attn_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)

        # MAS over the Gaussian log-likelihood of every (token, frame) pair (tts.py:130-139)
        with torch.no_grad():
            const = -0.5 * math.log(2 * math.pi) * self.n_feats
            factor = -0.5 * torch.ones(mu_x.shape, dtype=mu_x.dtype, device=mu_x.device)
            y_square = torch.matmul(factor.transpose(1, 2), y ** 2)
            y_mu_double = torch.matmul(2.0 * (factor * mu_x).transpose(1, 2), y)
            mu_square = torch.sum(factor * (mu_x ** 2), 1).unsqueeze(-1)
            log_prior = y_square - y_mu_double + mu_square + const
            attn = monotonic_align.maximum_path(log_prior, attn_mask.squeeze(1)).detach()

        logw_ = torch.log(1e-8 + torch.sum(attn.unsqueeze(1), -1)) * x_mask
        dur_loss = duration_loss(logw, logw_, x_lengths)

        if out_size is not None:          # random crop to out_size frames per item (tts.py:146-168)
            max_offset = (y_lengths - out_size).clamp(0)
            starts = [random.choice(range(0, int(e))) if int(e) > 0 else 0 for e in max_offset.cpu().numpy()]
            out_offset = torch.LongTensor(starts).to(y_lengths)
            attn_cut = torch.zeros(attn.shape[0], attn.shape[1], out_size, dtype=attn.dtype, device=attn.device)
            y_cut = torch.zeros(y.shape[0], self.n_feats, out_size, dtype=y.dtype, device=y.device)
            y_cut_lengths = []
            for i, (y_, off) in enumerate(zip(y, out_offset)):
                n = out_size + (y_lengths[i] - out_size).clamp(None, 0)
                y_cut_lengths.append(n)
                y_cut[i, :, :n] = y_[:, off:off + n]
                attn_cut[i, :, :n] = attn[i, :, off:off + n]
            y_cut_lengths = torch.LongTensor(y_cut_lengths)
            attn, y = attn_cut, y_cut
            y_mask = sequence_mask(y_cut_lengths).unsqueeze(1).to(y_mask)

        mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
        diff_loss, _ = self.decoder.compute_loss(y, y_mask, mu_y, spk)
        prior_loss = torch.sum(0.5 * ((y - mu_y) ** 2 + math.log(2 * math.pi)) * y_mask)
        prior_loss = prior_loss / (torch.sum(y_mask) * self.n_feats)
        return dur_loss, prior_loss, diff_loss
