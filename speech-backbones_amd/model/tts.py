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

    @staticmethod
    def _mas_scores(mu_x, y):
        """log N(y_j; mu_x_i, I) for every (token, frame) pair -> [B, t_x, T]   (tts.py:130-139).

        HIP tensors: one launch of gtts_log_prior, handed to MAS without leaving the device.  Host tensors (CPU-side
        training / tests): the same quantity from three reductions."""
        if mu_x.is_cuda:
            return backend().log_prior(mu_x, y)
        n_feats = mu_x.shape[1]
        energy_y = y.pow(2).sum(1)                                   # [B, T]
        energy_mu = mu_x.pow(2).sum(1)                               # [B, t_x]
        cross = torch.einsum("bfi,bfj->bij", mu_x, y)                # [B, t_x, T]
        return cross - 0.5 * energy_y.unsqueeze(1) - 0.5 * energy_mu.unsqueeze(2) - 0.5 * math.log(2 * math.pi) * n_feats

    def compute_loss(self, x, x_lengths, y, y_lengths, spk=None, out_size=None):
        """(duration loss, prior loss, diffusion loss)   (tts.py:101-181).

        x [B,t_x] token ids, y [B,F,T] target mels; out_size: frames of the random training crop (None = no crop)."""
        x, x_lengths, y, y_lengths = self.relocate_input([x, x_lengths, y, y_lengths])
        if self.n_spks > 1:
            spk = self.spk_emb(spk)
        mu_x, logw, x_mask = self.encoder(x, x_lengths, spk)
        frames = y.shape[-1]
        y_mask = sequence_mask(y_lengths, frames).unsqueeze(1).to(x_mask)
        pair_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)       # [B, 1, t_x, T]

        # hard alignment: most likely monotonic path through the (token, frame) log-likelihood matrix
        with torch.no_grad():
            attn = monotonic_align.maximum_path(self._mas_scores(mu_x, y), pair_mask.squeeze(1)).detach()

        # duration predictor target = log of the frames MAS gave each token (tts.py:141-143)
        frames_per_token = attn.sum(-1).unsqueeze(1)
        dur_loss = duration_loss(logw, torch.log(1e-8 + frames_per_token) * x_mask, x_lengths)

        if out_size is not None:
            # random out_size-frame window per utterance (tts.py:146-168), taken with one gather instead of a loop;
            # the window starts consume Python's `random` stream exactly like the reference's random.choice(range(n))
            slack = (y_lengths - out_size).clamp(min=0).tolist()
            starts = torch.tensor([random.randrange(int(n)) if int(n) > 0 else 0 for n in slack], dtype=torch.long,
                                  device=y.device)
            kept = torch.clamp(y_lengths, max=out_size)
            col = torch.arange(out_size, device=y.device)
            src = (starts.unsqueeze(1) + col.unsqueeze(0)).clamp(max=frames - 1)        # [B, out_size]
            live = (col.unsqueeze(0) < kept.unsqueeze(1)).unsqueeze(1)                   # [B, 1, out_size]
            y = torch.gather(y, 2, src.unsqueeze(1).expand(-1, y.shape[1], -1)) * live.to(y.dtype)
            attn = torch.gather(attn, 2, src.unsqueeze(1).expand(-1, attn.shape[1], -1)) * live.to(attn.dtype)
            y_mask = sequence_mask(kept).unsqueeze(1).to(y_mask)

        # aligned prior mean, diffusion loss on the decoder, Gaussian prior loss of the encoder (tts.py:171-181)
        mu_y = torch.matmul(attn.transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
        diff_loss, _ = self.decoder.compute_loss(y, y_mask, mu_y, spk)
        nll = 0.5 * ((y - mu_y) ** 2 + math.log(2 * math.pi)) * y_mask
        prior_loss = nll.sum() / (y_mask.sum() * self.n_feats)
        return dur_loss, prior_loss, diff_loss
