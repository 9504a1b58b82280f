"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- functional CPU restatement of DiffVC's PostNet
(DiffVC/model/postnet.py:15-53) over a plain state_dict; stock torch CPU fp32 ops are the reference's arithmetic.
Pinned by tests/golden/postnet.npz (output of the reference module, tests/golden/make_golden_postnet.py) and live in
tests/test_encoder_cpu.py."""
import math

import torch
import torch.nn.functional as F


def mish(x):
    """DiffVC/model/modules.py Mish: x * tanh(softplus(x))."""
    return x * torch.tanh(F.softplus(x))


def block(sd, p, x, mask, groups=8):
    """Block.forward, postnet.py:21-23."""
    y = F.conv2d(x * mask, sd[p + "block.0.weight"], sd[p + "block.0.bias"], padding=3)
    y = F.group_norm(y, groups, sd[p + "block.1.weight"], sd[p + "block.1.bias"], eps=1e-5)
    return mish(y) * mask


def postnet_forward(sd, x, mask, groups=8):
    """PostNet.forward, postnet.py:47-53 (ResnetBlock.forward :33-37).  x [B,F,T], mask [B,1,T]."""
    x = x.unsqueeze(1)
    m = mask.unsqueeze(1)
    x = F.conv2d(x * m, sd["init_conv.weight"], sd["init_conv.bias"])
    h = block(sd, "res_block.block1.", x, m, groups)
    h = block(sd, "res_block.block2.", h, m, groups)
    x = F.conv2d(x * m, sd["res_block.res.weight"], sd["res_block.res.bias"]) + h
    return F.conv2d(x * m, sd["final_conv.weight"], sd["final_conv.bias"]).squeeze(1)


def make_state(dim=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = uni((cout, cin, k, k), cin * k * k)
        sd[name + ".bias"] = uni((cout,), cin * k * k)

    conv("init_conv", dim, 1, 1)
    for b in ("block1", "block2"):
        conv("res_block.%s.block.0" % b, dim, dim, 7)
        sd["res_block.%s.block.1.weight" % b] = 1.0 + 0.2 * (torch.rand(dim, generator=g) - 0.5)
        sd["res_block.%s.block.1.bias" % b] = 0.2 * (torch.rand(dim, generator=g) - 0.5)
    conv("res_block.res", dim, dim, 1)
    conv("final_conv", 1, dim, 1)
    return sd
