"""PostNet -- drop-in for DiffVC/model/postnet.py:15-53 (same module tree and state_dict: `init_conv`, `res_block.block{1,2}.
block.{0,1}`, `res_block.res`, `final_conv`).  Inference on HIP tensors runs gtts_postnet_forward (csrc/postnet.hip: the two
7x7 convolutions on the MFMA kernel with GroupNorm + Mish applied on load); with autograd it composes torch ops."""
import torch
import torch.nn.functional as F

from .base import BaseModule


def _mish(v):
    return v * torch.tanh(F.softplus(v))


class Block(BaseModule):
    """postnet.py:15-23: Conv2d 7x7 on x * mask -> GroupNorm -> Mish, * mask."""

    def __init__(self, dim, groups=8):
        super().__init__()
        self.block = torch.nn.Sequential(torch.nn.Conv2d(dim, dim, 7, padding=3), torch.nn.GroupNorm(groups, dim),
                                         torch.nn.Identity())     # index 2 is the parameter-free Mish of the reference

    def forward(self, x, mask):
        y = self.block[1](self.block[0](x * mask))
        return _mish(y) * mask


class ResnetBlock(BaseModule):
    """postnet.py:26-37."""

    def __init__(self, dim, groups=8):
        super().__init__()
        self.block1 = Block(dim, groups=groups)
        self.block2 = Block(dim, groups=groups)
        self.res = torch.nn.Conv2d(dim, dim, 1)

    def forward(self, x, mask):
        return self.res(x * mask) + self.block2(self.block1(x, mask), mask)


class PostNet(BaseModule):
    """postnet.py:40-53."""

    def __init__(self, dim, groups=8):
        super().__init__()
        self.dim, self.groups = dim, groups
        self.init_conv = torch.nn.Conv2d(1, dim, 1)
        self.res_block = ResnetBlock(dim, groups=groups)
        self.final_conv = torch.nn.Conv2d(dim, 1, 1)
        self._hip = None
        self._hip_blob = None
        self._hip_key = None

    def invalidate_packed(self):
        self._hip_blob = None
        self._hip_key = None

    def forward(self, x, mask):
        """x [B, n_feats, T], mask [B, 1, T] -> [B, n_feats, T]."""
        if x.is_cuda and not torch.is_grad_enabled():
            from ...model._backend import backend
            if self._hip is None:
                self._hip = backend().PostNetPlan(self.dim, x.shape[1], self.groups)
            params = list(self.named_parameters())
            key = (str(x.device),) + tuple((p.data_ptr(), p._version) for _, p in params)
            if self._hip_blob is None or self._hip_key != key:
                self._hip_blob = self._hip.pack({n: p for n, p in params}, x.device)
                self._hip_key = key
            return self._hip.forward(self._hip_blob, x, mask)
        v, m = x.unsqueeze(1), mask.unsqueeze(1)
        v = self.init_conv(v * m)
        v = self.res_block(v, m)
        return self.final_conv(v * m).squeeze(1)
