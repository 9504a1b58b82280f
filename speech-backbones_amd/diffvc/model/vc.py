"""FwdDiffusion and DiffVC -- the model shell of DiffVC/model/vc.py:17-127 (constructor arguments, attribute names, module
tree and therefore state_dict keys: `encoder.encoder.*`, `encoder.postnet.*`, `decoder.estimator.*`).

Inference on HIP tensors composes the three C-ABI paths of this package: MelEncoder (gtts_enc_forward, mode 1), PostNet
(gtts_postnet_forward) and the decoder's sampler (gtts_vc_reverse_diffusion); the speaker encoder that produces `c` stays
the caller's (DiffVC/inference.ipynb loads it separately).  Training methods compose the modules' autograd paths."""
import torch

from .base import BaseModule
from .diffusion import Diffusion
from .encoder import MelEncoder
from .postnet import PostNet
from .utils import fix_len_compatibility, mse_loss, sequence_mask


class FwdDiffusion(BaseModule):
    """The "average voice" encoder that parameterises the diffusion prior (vc.py:17-49)."""

    def __init__(self, n_feats, channels, filters, heads, layers, kernel, dropout, window_size, dim):
        super().__init__()
        self.n_feats, self.channels, self.filters, self.heads = n_feats, channels, filters, heads
        self.layers, self.kernel, self.dropout, self.window_size, self.dim = layers, kernel, dropout, window_size, dim
        self.encoder = MelEncoder(n_feats, channels, filters, heads, layers, kernel, dropout, window_size)
        self.postnet = PostNet(dim)

    def _average_voice(self, x, mask):
        return self.postnet(self.encoder(x, mask), mask)

    @torch.no_grad()
    def forward(self, x, mask):
        x, mask = self.relocate_input([x, mask])
        return self._average_voice(x, mask)

    def compute_loss(self, x, y, mask):
        x, y, mask = self.relocate_input([x, y, mask])
        return mse_loss(self._average_voice(x, mask), y, mask, self.n_feats)


class DiffVC(BaseModule):
    """Average-voice encoder + speaker-conditional diffusion decoder (vc.py:52-148)."""

    def __init__(self, n_feats, channels, filters, heads, layers, kernel, dropout, window_size, enc_dim, spk_dim, use_ref_t,
                 dec_dim, beta_min, beta_max):
        super().__init__()
        self.n_feats, self.channels, self.filters, self.heads = n_feats, channels, filters, heads
        self.layers, self.kernel, self.dropout, self.window_size = layers, kernel, dropout, window_size
        self.enc_dim, self.spk_dim, self.use_ref_t, self.dec_dim = enc_dim, spk_dim, use_ref_t, dec_dim
        self.beta_min, self.beta_max = beta_min, beta_max
        self.encoder = FwdDiffusion(n_feats, channels, filters, heads, layers, kernel, dropout, window_size, enc_dim)
        self.decoder = Diffusion(n_feats, dec_dim, spk_dim, use_ref_t, beta_min, beta_max)

    def load_encoder(self, enc_path):
        self.encoder.load_state_dict(torch.load(enc_path, map_location="cpu"), strict=False)

    @torch.no_grad()
    def forward(self, x, x_lengths, x_ref, x_ref_lengths, c, n_timesteps, mode="ml"):
        """Source mels x [B, F, T] (lengths x_lengths), reference mels x_ref, speaker embeddings c -> (diffused average voice
        of the source, converted mels cut to the longest source).  mode: 'pf' | 'em' | 'ml' (vc.py:82-127)."""
        x, x_lengths = self.relocate_input([x, x_lengths])
        x_ref, x_ref_lengths, c = self.relocate_input([x_ref, x_ref_lengths, c])
        x_mask = sequence_mask(x_lengths).unsqueeze(1).to(x.dtype)
        ref_mask = sequence_mask(x_ref_lengths).unsqueeze(1).to(x_ref.dtype)
        mean = self.encoder(x, x_mask)
        mean_x = self.decoder.compute_diffused_mean(x, x_mask, mean, 1.0)
        mean_ref = self.encoder(x_ref, ref_mask)
        # the decoder halves the frame axis twice: pad to the next multiple of four, valid frames copied, the rest zero
        t_max = int(x_lengths.max())
        t_pad = fix_len_compatibility(t_max)
        mask_pad = sequence_mask(x_lengths, t_pad).unsqueeze(1).to(x.dtype)
        keep = mask_pad[:, :, :t_max] * sequence_mask(x_lengths, t_max).unsqueeze(1).to(x.dtype)
        mean_pad = x.new_zeros(x.shape[0], self.n_feats, t_pad)
        z = x.new_zeros(x.shape[0], self.n_feats, t_pad)
        mean_pad[:, :, :t_max] = mean[:, :, :t_max] * keep
        z[:, :, :t_max] = mean_x[:, :, :t_max] * keep
        z += torch.randn_like(z)
        y = self.decoder(z, mask_pad, mean_pad, x_ref, ref_mask, mean_ref, c, n_timesteps, mode)
        return mean_x, y[:, :, :t_max]

    def compute_loss(self, x, x_lengths, x_ref, c):
        """Score-matching loss of the decoder with the (detached) average-voice encoder outputs (vc.py:129-148)."""
        x, x_lengths, x_ref, c = self.relocate_input([x, x_lengths, x_ref, c])
        x_mask = sequence_mask(x_lengths).unsqueeze(1).to(x.dtype)
        mean = self.encoder(x, x_mask).detach()
        mean_ref = self.encoder(x_ref, x_mask).detach()
        return self.decoder.compute_loss(x, x_mask, mean, x_ref, mean_ref, c)
