"""CPU tests (no GPU) of the algebraic identities the training kernels rely on (csrc/train*.hip, model/_train_ops.py): each
gradient that is computed by re-using another kernel is checked here with stock torch ops against torch autograd, so that a
GPU parity failure can be told apart from a wrong identity.  Shapes are small; everything is exact up to fp32 rounding."""
import importlib

import pytest
import torch
import torch.nn.functional as F

T = importlib.import_module("speech-backbones_amd.model._train_ops")


def _close(a, b, tol=2e-5):
    return float((a - b).abs().max()) <= tol * (1.0 + float(b.abs().max()))


def test_downsample_data_gradient_is_an_upsample_with_zero_padded_kernel():
    """d/dx of Conv2d(3x3, stride 2, pad 1) == ConvTranspose2d(4x4, stride 2, pad 1) of dy with the forward weight zero-padded
    to 4x4 (ResampleConv.backward, gtts_conv_resample 'dn_T'): same index map y = 2 oy - 1 + ky."""
    torch.manual_seed(0)
    x = torch.randn(2, 3, 8, 12, requires_grad=True)
    w = torch.randn(5, 3, 3, 3)
    y = F.conv2d(x, w, None, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    got = F.conv_transpose2d(dy, F.pad(w, (0, 1, 0, 1)), None, 2, 1)
    assert got.shape == x.shape and _close(got, x.grad)


def test_downsample_weight_gradient_is_the_stride1_one_against_zero_inserted_dy():
    """dW of the stride-2 convolution == dW of the stride-1 3x3 convolution whose output gradient is dy at the even positions and
    zero elsewhere (gtts_zero_insert2 + gtts_conv3x3_wgrad_tiled)."""
    torch.manual_seed(1)
    x = torch.randn(2, 3, 8, 12)
    w = torch.randn(5, 3, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    dyz = torch.zeros(2, 5, 8, 12)
    dyz[:, :, ::2, ::2] = dy
    w1 = torch.zeros_like(w, requires_grad=True)
    F.conv2d(x, w1, None, 1, 1).backward(dyz)
    assert _close(w1.grad, w.grad)


def test_upsample_gradients_over_the_four_phases_of_dy():
    """ConvTranspose2d(4x4, stride 2, pad 1): both gradients as 3x3 stride-1 operations over the four stride-2 phases of dy
    stacked as channels (gtts_space_to_depth2; ResampleConv.backward): odd rows meet kernel rows 0, 2 at offsets -1, 0; even rows
    meet rows 1, 3 at offsets 0, +1 (the same for columns)."""
    torch.manual_seed(2)
    B, ci, co, h, w = 2, 3, 5, 4, 6
    x = torch.randn(B, ci, h, w, requires_grad=True)
    W = torch.randn(ci, co, 4, 4, requires_grad=True)
    y = F.conv_transpose2d(x, W, None, 2, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    P = torch.empty(B, 4 * co, h, w)                  # block (pr * 2 + pc): rows 2y + 1 - pr, columns 2x + 1 - pc
    for pr in (0, 1):
        for pc in (0, 1):
            P[:, (pr * 2 + pc) * co:(pr * 2 + pc + 1) * co] = dy[:, :, (1 - pr)::2, (1 - pc)::2]
    tap = T._up_tap_index(torch.device("cpu"))
    wz = torch.cat((W.detach(), torch.zeros(ci, co, 1, 4)), 2)
    wz = torch.cat((wz, torch.zeros(ci, co, 5, 1)), 3)
    wp = wz[:, :, tap[:, None, :, None], tap[None, :, None, :]].permute(0, 2, 3, 1, 4, 5).reshape(ci, 4 * co, 3, 3)
    assert _close(F.conv2d(P, wp, padding=1), x.grad)
    wd = torch.zeros(ci, 4 * co, 3, 3, requires_grad=True)         # "weight gradient" with the operands' roles swapped
    F.conv2d(P, wd, padding=1).backward(x.detach())
    pr, kp = T._up_tap_inverse(torch.device("cpu"))
    d6 = wd.grad.reshape(ci, 2, 2, co, 3, 3)
    gw = d6[:, pr[:, None], pr[None, :], :, kp[:, None], kp[None, :]].permute(2, 3, 0, 1)
    assert _close(gw, W.grad)


def test_linear_attention_backward_formulas():
    """The closed forms of csrc/train_attn.hip -- in particular the softmax row term r_d = sum_e dctx[d][e] ctx[d][e], which needs
    no second pass over the pixels -- against autograd of the reference's composition (diffusion.py:90-100)."""
    torch.manual_seed(3)
    n = 37
    q, k, v = (torch.randn(32, n, requires_grad=True) for _ in range(3))
    kt = torch.softmax(k, dim=-1)
    ctx = kt @ v.t()                                  # [d][e]
    out = ctx.t() @ q                                 # [e][n]
    dout = torch.randn_like(out)
    out.backward(dout)
    ktd, ctxd = kt.detach(), ctx.detach()
    dq = ctxd @ dout
    dctx = q.detach() @ dout.t()
    dv = dctx.t() @ ktd
    r = (dctx * ctxd).sum(1, keepdim=True)
    dk = ktd * (dctx @ v.detach() - r)
    assert _close(dq, q.grad) and _close(dv, v.grad) and _close(dk, k.grad, 1e-4)


def test_one_by_one_data_gradient_mask_commutes():
    """A 1x1 convolution does not mix columns: masking dy's columns before the transposed convolution equals masking dx after
    it (MaskedConv1x1.backward hands the mask to the kernel's input prologue)."""
    torch.manual_seed(4)
    w = torch.randn(6, 4, 1, 1)
    dy = torch.randn(2, 6, 5, 7)
    m = (torch.rand(2, 1, 1, 7) > 0.4).float()
    a = F.conv_transpose2d(dy * m, w)
    b = F.conv_transpose2d(dy, w) * m
    assert torch.equal(a, b)


def test_time_terms_equal_the_per_block_mlps():
    """ResnetBlock.mlp of every block in one pass (time_terms) == the reference's per-block Mish -> Linear (diffusion.py:66-67)."""
    D = importlib.import_module("speech-backbones_amd.model.diffusion")
    torch.manual_seed(5)
    blocks = [D.ResnetBlock(8, c, time_emb_dim=16) for c in (8, 16, 24)]
    temb = torch.randn(3, 16)
    got = T.time_terms(blocks, temb)
    for rb, g in zip(blocks, got):
        assert _close(g, rb.mlp(temb), 1e-6)
