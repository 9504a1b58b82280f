"""MelEncoder -- drop-in for DiffVC/model/encoder.py:257-284 (the "average voice" encoder that feeds the DiffVC decoder):
init_proj (1x1) -> ConvReluNorm prenet -> relative-position transformer Encoder -> term_proj (1x1).

The prenet / Encoder building blocks are the Grad-TTS text-encoder modules (the two reference files are the same
glow-tts code); inference on HIP tensors runs the kernels of csrc/enc.hip (gtts_enc_forward, mode 1).
"""
import torch

from ...model.text_encoder import ConvReluNorm, Encoder
from .base import BaseModule


class MelEncoder(BaseModule):
    def __init__(self, n_feats, channels, filters, heads, layers, kernel, dropout, window_size=None):
        super().__init__()
        self.n_feats, self.channels, self.filters, self.heads = n_feats, channels, filters, heads
        self.layers, self.kernel, self.dropout, self.window_size = layers, kernel, dropout, window_size
        self.init_proj = torch.nn.Conv1d(n_feats, channels, 1)
        self.prenet = ConvReluNorm(channels, channels, channels, kernel_size=5, n_layers=3, p_dropout=0.5)
        self.encoder = Encoder(channels, filters, heads, layers, kernel, dropout, window_size=window_size)
        self.term_proj = torch.nn.Conv1d(channels, n_feats, 1)
        self._hip_enc = None
        self._hip_blob = None
        self._hip_key = None

    def invalidate_packed(self):
        self._hip_blob = None
        self._hip_key = None

    def forward(self, x, x_mask):
        """x [B, n_feats, T], x_mask [B, 1, T] -> [B, n_feats, T]   (encoder.py:279-284)."""
        if x.is_cuda and not torch.is_grad_enabled() and not self.training:
            from ...model._backend import backend
            be = backend()
            if self._hip_enc is None:
                self._hip_enc = be.Encoder("mel", 0, self.n_feats, self.channels, self.filters, 0, self.heads, self.layers,
                                           self.kernel, self.window_size)
            params = list(self.named_parameters())
            key = (str(x.device),) + tuple((p.data_ptr(), p._version) for _, p in params)
            if self._hip_blob is None or self._hip_key != key:
                self._hip_blob = self._hip_enc.pack({n: p for n, p in params}, x.device)
                self._hip_key = key
            return self._hip_enc.forward(self._hip_blob, x, x_mask)
        x = self.init_proj(x * x_mask)
        x = self.prenet(x, x_mask)
        x = self.encoder(x, x_mask)
        return self.term_proj(x * x_mask)
