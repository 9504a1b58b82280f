"""HiFi-GAN Generator -- drop-in for Grad-TTS/hifi-gan/models.py:13-128 (same constructor `Generator(h)`, same
parameter names / shapes with and without weight normalisation, same `remove_weight_norm()`), sampling on the MI355X
HIP kernels (csrc/voc.hip) behind the C ABI.

`forward(x)` on a HIP tensor under torch.no_grad() runs gtts_voc_forward: one 1-D MFMA convolution kernel for every
Conv1d / ConvTranspose1d with LeakyReLU-on-load, residual and ResBlock-mean epilogues.  With autograd enabled (GAN
training, out of this repo's scope) it composes stock torch ops over the same parameters.
"""
import torch
import torch.nn.functional as F
from torch.nn import Conv1d, ConvTranspose1d
from torch.nn.utils import remove_weight_norm, weight_norm

try:
    from .xutils import get_padding, init_weights
except ImportError:          # imported as top-level `models` (how inference.py does it)
    from xutils import get_padding, init_weights

LRELU_SLOPE = 0.1


def _backend():
    import importlib.util
    import os
    import sys
    try:
        from .. import _lib
        return _lib
    except (ImportError, ValueError):
        name = "gradtts_mi355x_lib"
        if name not in sys.modules:
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_lib.py")
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
        return sys.modules[name]


def _wn_conv(channels, k, dilation):
    return weight_norm(Conv1d(channels, channels, k, 1, dilation=dilation, padding=get_padding(k, dilation)))


class ResBlock1(torch.nn.Module):
    """models.py:13-50: three (dilated conv, conv) pairs, each with a residual connection."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.convs1 = torch.nn.ModuleList([_wn_conv(channels, kernel_size, d) for d in dilation])
        self.convs1.apply(init_weights)
        self.convs2 = torch.nn.ModuleList([_wn_conv(channels, kernel_size, 1) for _ in dilation])
        self.convs2.apply(init_weights)

    def forward(self, x):
        for first, second in zip(self.convs1, self.convs2):
            x = second(F.leaky_relu(first(F.leaky_relu(x, LRELU_SLOPE)), LRELU_SLOPE)) + x
        return x

    def remove_weight_norm(self):
        for conv in list(self.convs1) + list(self.convs2):
            remove_weight_norm(conv)


class ResBlock2(torch.nn.Module):
    """models.py:53-74: two dilated convs, each with a residual connection."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.h = h
        self.convs = torch.nn.ModuleList([_wn_conv(channels, kernel_size, d) for d in dilation])
        self.convs.apply(init_weights)

    def forward(self, x):
        for conv in self.convs:
            x = conv(F.leaky_relu(x, LRELU_SLOPE)) + x
        return x

    def remove_weight_norm(self):
        for conv in self.convs:
            remove_weight_norm(conv)


class Generator(torch.nn.Module):
    """models.py:77-128."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        width = h.upsample_initial_channel
        self.conv_pre = weight_norm(Conv1d(80, width, 7, 1, padding=3))
        block = ResBlock1 if h.resblock == '1' else ResBlock2
        self.ups = torch.nn.ModuleList()
        for i, (rate, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(weight_norm(ConvTranspose1d(width // 2 ** i, width // 2 ** (i + 1), k, rate,
                                                        padding=(k - rate) // 2)))
        self.resblocks = torch.nn.ModuleList()
        for i in range(len(self.ups)):
            ch = width // 2 ** (i + 1)
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(block(h, ch, k, d))
        self.conv_post = weight_norm(Conv1d(ch, 1, 7, 1, padding=3))
        self.ups.apply(init_weights)
        self.conv_post.apply(init_weights)
        self._voc = None
        self._voc_blob = None
        self._voc_key = None

    # ---- HIP plumbing ------------------------------------------------------------------------------
    def _effective_state(self):
        """name -> effective weight / bias (weight normalisation folded when it is still attached)."""
        out = {}
        for name, mod in self.named_modules():
            if isinstance(mod, (Conv1d, ConvTranspose1d)):
                out[name + ".weight"] = mod.weight.detach()       # the weight_norm pre-hook refreshed .weight at the last
                out[name + ".bias"] = mod.bias.detach()           # forward; recompute it here for safety
                if hasattr(mod, "weight_g"):
                    out[name + ".weight"] = torch._weight_norm(mod.weight_v.detach(), mod.weight_g.detach(), 0)
        return out

    def invalidate_packed(self):
        self._voc_blob = None
        self._voc_key = None

    def _packed(self, device):
        be = _backend()
        if self._voc is None:
            self._voc = be.Vocoder.from_config(self.h)
        params = list(self.parameters())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if self._voc_blob is None or self._voc_key != key:
            self._voc_blob = self._voc.pack(self._effective_state(), device)
            self._voc_key = key
        return self._voc_blob

    def _forward_torch(self, x):
        x = self.conv_pre(x)
        for i, up in enumerate(self.ups):
            x = up(F.leaky_relu(x, LRELU_SLOPE))
            branches = self.resblocks[i * self.num_kernels:(i + 1) * self.num_kernels]
            total = branches[0](x)
            for blk in branches[1:]:
                total = total + blk(x)
            x = total / self.num_kernels
        return torch.tanh(self.conv_post(F.leaky_relu(x)))

    def forward(self, x):
        """mel [B, 80, T] -> waveform [B, 1, 256 T]."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_torch(x)
        if not x.is_cuda:
            raise RuntimeError("HiFi-GAN Generator inference runs on the MI355X HIP kernels only; got a %s tensor "
                               "(there is no CPU fallback)" % x.device)
        return self._voc_forward(x)

    def _voc_forward(self, x):
        blob = self._packed(x.device)
        return self._voc.forward(blob, x)

    def remove_weight_norm(self):
        print('Removing weight norm...')
        for up in self.ups:
            remove_weight_norm(up)
        for blk in self.resblocks:
            blk.remove_weight_norm()
        remove_weight_norm(self.conv_pre)
        remove_weight_norm(self.conv_post)
        self.invalidate_packed()
