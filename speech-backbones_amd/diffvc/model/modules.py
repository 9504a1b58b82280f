"""Building blocks of the DiffVC decoder (DiffVC/model/modules.py).  Lines 16-110 of the reference are byte-identical
to Grad-TTS's blocks, so those classes are shared; SinusoidalPosEmb hard-codes the 1000x scale (modules.py:123) and
RefBlock (modules.py:128-166) is specific to DiffVC."""
import math

import torch

from ...model.base import BaseModule
from ...model.diffusion import (Block, Downsample, LinearAttention, Mish, Residual, ResnetBlock, Rezero,  # noqa: F401
                                Upsample)


class SinusoidalPosEmb(BaseModule):
    """modules.py:113-125."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device).float() * -(math.log(10000) / (half - 1)))
        arg = 1000.0 * x[:, None] * freq[None, :]
        return torch.cat((arg.sin(), arg.cos()), dim=-1)


def _conv_in_glu(cin, cout):
    return torch.nn.Sequential(torch.nn.Conv2d(cin, cout, 3, 1, 1), torch.nn.InstanceNorm2d(cout, affine=True),
                               torch.nn.GLU(dim=1))


class RefBlock(BaseModule):
    """Reference-mel summariser: 6 x (conv3x3 -> InstanceNorm -> GLU) with two time-bias adds, 1x1 conv, masked mean."""

    def __init__(self, out_dim, time_emb_dim):
        super().__init__()
        base = out_dim // 4
        self.mlp1 = torch.nn.Sequential(Mish(), torch.nn.Linear(time_emb_dim, base))
        self.mlp2 = torch.nn.Sequential(Mish(), torch.nn.Linear(time_emb_dim, 2 * base))
        self.block11 = _conv_in_glu(1, 2 * base)
        self.block12 = _conv_in_glu(base, 2 * base)
        self.block21 = _conv_in_glu(base, 4 * base)
        self.block22 = _conv_in_glu(2 * base, 4 * base)
        self.block31 = _conv_in_glu(2 * base, 8 * base)
        self.block32 = _conv_in_glu(4 * base, 8 * base)
        self.final_conv = torch.nn.Conv2d(4 * base, out_dim, 1)

    @staticmethod
    def _block(blk, y, mask):
        """conv3x3(y * mask) -> InstanceNorm -> GLU (DiffVC/model/modules.py:140-157).  Training on the GPU: the convolution (forward, data
        and weight gradient) on the gtts:: training kernels where their tiles allow (64 or a multiple of 128 channels on both sides:
        block22 / block31 / block32 = 90 % of RefBlock's FLOPs at out_dim 128; the three narrow layers stay on the framework's
        convolution), InstanceNorm + GLU of all six blocks on csrc/train_inglu.hip."""
        from ...model import _train_ops as T
        from ...model._backend import backend
        conv, norm = blk[0], blk[1]
        if not (torch.is_grad_enabled() and T._hip(y) and y.dim() == 4 and not T.FORCE_TORCH):
            return blk(y * mask)
        if backend().conv3x3_supported(conv.in_channels, conv.out_channels, need_dgrad=True, shape=(y.shape[0], y.shape[2], y.shape[3])):
            T._count(True)
            h = T.MaskedConv3x3.apply(y.contiguous(), mask, conv.weight, conv.bias, None)
        else:
            h = conv(y * mask)
        if norm.affine and not norm.track_running_stats:
            T._count(True)
            return T.InstNormGlu.apply(h.contiguous(), norm.weight, norm.bias, norm.eps)
        return blk[2](norm(h))

    def forward(self, x, mask, time_emb):
        y = self._block(self.block12, self._block(self.block11, x, mask), mask)
        y = y + self.mlp1(time_emb)[:, :, None, None]
        y = self._block(self.block22, self._block(self.block21, y, mask), mask)
        y = y + self.mlp2(time_emb)[:, :, None, None]
        y = self._block(self.block32, self._block(self.block31, y, mask), mask)
        y = self.final_conv(y * mask)
        return (y * mask).sum((2, 3)) / (mask.sum((2, 3)) * x.shape[2])
