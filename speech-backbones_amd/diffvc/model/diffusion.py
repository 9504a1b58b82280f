"""Drop-in mirror of DiffVC/model/diffusion.py for MI355X: `GradLogPEstimator` (speaker-conditional score U-Net,
diffusion.py:17-106) and `Diffusion` with the pf / em / ml samplers (diffusion.py:109-222).  Same constructor
signatures, parameter names and shapes (117 794 599 decoder parameters at dim_unet=256).

Sampling (torch.no_grad) runs on the HIP kernels through the C ABI (gtts_vc_*); there is no CPU fallback.  Training
(loss_t / compute_loss, autograd) runs the score network's trunk on the gtts:: training kernels (GradLogPEstimator._forward_train).
"""
import math

import torch

from ...model import _train_ops as T
from ...model._backend import backend
from ...model.base import BaseModule
from .modules import (Block, Downsample, LinearAttention, Mish, RefBlock, Residual, ResnetBlock, Rezero,
                      SinusoidalPosEmb, Upsample)


class GradLogPEstimator(BaseModule):
    def __init__(self, dim_base, dim_cond, use_ref_t, dim_mults=(1, 2, 4)):
        super().__init__()
        self.use_ref_t = use_ref_t
        self.dim_base, self.dim_cond, self.dim_mults = dim_base, dim_cond, dim_mults
        widths = [2 + dim_cond] + [dim_base * m for m in dim_mults]
        stages = list(zip(widths[:-1], widths[1:]))
        self.time_pos_emb = SinusoidalPosEmb(dim_base)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(dim_base, 4 * dim_base), Mish(),
                                       torch.nn.Linear(4 * dim_base, dim_base))
        cond_total = dim_base + 256
        if use_ref_t:
            self.ref_block = RefBlock(out_dim=dim_cond, time_emb_dim=dim_base)
            cond_total += dim_cond
        self.cond_block = torch.nn.Sequential(torch.nn.Linear(cond_total, 4 * dim_cond), Mish(),
                                              torch.nn.Linear(4 * dim_cond, dim_cond))
        self.downs = torch.nn.ModuleList()
        self.ups = torch.nn.ModuleList()          # created before the mid blocks: fixes the registration order
        for i, (cin, cout) in enumerate(stages):
            last = i == len(stages) - 1
            self.downs.append(torch.nn.ModuleList([
                ResnetBlock(cin, cout, time_emb_dim=dim_base), ResnetBlock(cout, cout, time_emb_dim=dim_base),
                Residual(Rezero(LinearAttention(cout))), torch.nn.Identity() if last else Downsample(cout)]))
        mid = widths[-1]
        self.mid_block1 = ResnetBlock(mid, mid, time_emb_dim=dim_base)
        self.mid_attn = Residual(Rezero(LinearAttention(mid)))
        self.mid_block2 = ResnetBlock(mid, mid, time_emb_dim=dim_base)
        for cin, cout in reversed(stages[1:]):
            self.ups.append(torch.nn.ModuleList([
                ResnetBlock(2 * cout, cin, time_emb_dim=dim_base), ResnetBlock(cin, cin, time_emb_dim=dim_base),
                Residual(Rezero(LinearAttention(cin))), Upsample(cin)]))
        self.final_block = Block(dim_base, dim_base)
        self.final_conv = torch.nn.Conv2d(dim_base, 1, 1)
        self._beta_range = (0.05, 20.0)
        self._hip_plan = None
        self._hip_blob = None
        self._hip_key = None

    # ---- HIP plumbing
    def _plan(self):
        if tuple(self.dim_mults) != (1, 2, 4):
            raise RuntimeError("the HIP path supports dim_mults=(1,2,4) (the reference's configuration)")
        key = (float(self._beta_range[0]), float(self._beta_range[1]))
        if self._hip_plan is None or getattr(self, "_hip_plan_key", None) != key:
            self._hip_plan = backend().Plan(dim=self.dim_base, arch=1, dim_cond=self.dim_cond, use_ref_t=self.use_ref_t,
                                            c_dim=256, pe_scale=1000.0, beta_min=key[0], beta_max=key[1])
            self._hip_plan_key = key
            self.invalidate_packed()
        return self._hip_plan

    def invalidate_packed(self):
        """Drop the packed blob (needed after edits through `p.data`, which do not bump Tensor._version)."""
        self._hip_blob = None
        self._hip_key = None

    def _packed(self, device):
        plan = self._plan()
        params = list(self.named_parameters())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for _, p in params)
        if self._hip_blob is None or self._hip_key != key:
            self._hip_blob = plan.pack({n: p for n, p in params}, device)
            self._hip_key = key
        return self._hip_blob

    def _first_resnet(self, rb, v2, cexp, m, temb):
        """ResnetBlock.forward (DiffVC/model/modules.py:75-103) on cat(v2, cexp) -- the stacked (mean, x) planes and the condition
        vector broadcast over the mel plane, 2 + dim_cond channels (diffusion.py:74-76) -- without the concatenated tensor: both of
        its convolutions are linear in their input, so each is the sum of a 2-channel convolution (the first-layer kernels of the
        Grad-TTS network: no data gradient wanted) and a dim_cond-channel one with the weight's column blocks."""
        be = backend()
        conv, norm, rc = rb.block1.block[0], rb.block1.block[1], rb.res_conv
        nc, co = cexp.shape[1], conv.out_channels
        shape = (v2.shape[0], v2.shape[2], v2.shape[3])
        ok = (T._hip(v2) and not v2.requires_grad and
              be.conv3x3_supported(2, co, need_dgrad=False, shape=shape) and be.conv3x3_supported(nc, co, need_dgrad=True, shape=shape) and
              be.conv1x1_supported(2, co, need_dgrad=False, shape=shape) and be.conv1x1_supported(nc, co, need_dgrad=True, shape=shape))
        if not ok:
            T._count(False)
            return T.resnet(rb, torch.cat((v2, cexp), 1), m, temb)
        lin = rb.mlp[1]
        tb = torch.nn.functional.linear(T._mish(temb), lin.weight, lin.bias)
        T._count(True)
        y = T.MaskedConv3x3.apply(v2.contiguous(), m, conv.weight[:, :2].contiguous(), conv.bias, None)
        zero_b = torch.zeros(co, dtype=v2.dtype, device=v2.device)      # (the 3x3 entry point takes a bias pointer; the bias rides with the first part)
        y = T.MaskedResidualAdd.apply(y, T.MaskedConv3x3.apply(cexp, m, conv.weight[:, 2:].contiguous(), zero_b, None), None)
        T._count(True)
        h = T.GnMishMask.apply(y.contiguous(), m, norm.weight, norm.bias, norm.num_groups, norm.eps, tb.contiguous())
        h = T._conv_gn_mish(rb.block2, h, m)
        T._count(True)
        r = T.MaskedConv1x1.apply(v2.contiguous(), m, rc.weight[:, :2].contiguous(), rc.bias)
        r = T.MaskedResidualAdd.apply(r, T.MaskedConv1x1.apply(cexp, m, rc.weight[:, 2:].contiguous(), None), None)
        return T.MaskedResidualAdd.apply(h, r.contiguous(), None)

    def _forward_train(self, x, x_mask, mean, ref, ref_mask, c, t):
        """Autograd composition (training: DiffVC/train_dec.py:90-103 -> Diffusion.compute_loss -> loss_t, diffusion.py:207-226;
        this is GradLogPEstimator.forward, diffusion.py:61-106).  The trunk -- every 3x3 / 1x1 / resampling convolution with its
        data and weight gradients, GroupNorm + Mish + time term, the LinearAttention core, the residual adds, the final conv -- runs
        on the gtts:: training kernels of the Grad-TTS network (model/_train_ops.py; channel counts 64 or multiples of 128, i.e. every
        dim_base that is one); the condition path (time MLP, RefBlock's InstanceNorm + GLU stack on the one-channel reference mel,
        cond_block) composes stock torch ops: ~5 % of the step's FLOPs.  CPU tensors take the same code on stock ops."""
        if not T.FORCE_TORCH and x.is_cuda:
            be = backend()
            be.new_pack_generation()
            be.prepack(T._pack_specs(self))
        condition = self.time_pos_emb(t)
        temb = self.mlp(condition)
        v2 = torch.stack([mean, x], 1)
        m0 = x_mask.unsqueeze(1)
        if self.use_ref_t:
            condition = torch.cat([condition, self.ref_block(ref, ref_mask.unsqueeze(1), temb)], 1)
        condition = self.cond_block(torch.cat([condition, c], 1))
        cexp = condition[:, :, None, None].expand(-1, -1, v2.shape[2], v2.shape[3]).contiguous()
        skips, pyramid = [], [m0]
        first = True
        for r1, r2, att, down in self.downs:
            m = pyramid[-1]
            v = self._first_resnet(r1, v2, cexp, m, temb) if first else T.resnet(r1, v, m, temb)
            first = False
            v = T.attention(att, T.resnet(r2, v, m, temb))
            skips.append(v)
            if not isinstance(down, torch.nn.Identity):
                v = T._resample(v, m, down.conv, False)
            pyramid.append(m[..., ::2].contiguous())
        pyramid.pop()
        m = pyramid[-1]
        v = T.resnet(self.mid_block2, T.attention(self.mid_attn, T.resnet(self.mid_block1, v, m, temb)), m, temb)
        for r1, r2, att, up in self.ups:
            m = pyramid.pop()
            v = T.resnet(r1, v, m, temb, v1=skips.pop())            # (torch.cat((v, skip), 1) read in place)
            v = T.attention(att, T.resnet(r2, v, m, temb))
            v = T._resample(v, m, up.conv, True)
        v = T._conv_gn_mish(self.final_block, v, m0)
        return T.final_conv(self.final_conv, v, m0)

    def forward(self, x, x_mask, mean, ref, ref_mask, c, t):
        if torch.is_grad_enabled():
            return self._forward_train(x, x_mask, mean, ref, ref_mask, c, t)
        if not x.is_cuda:
            raise RuntimeError("GradLogPEstimator sampling runs on the MI355X HIP kernels only; got a %s tensor "
                               "(there is no CPU fallback)" % x.device)
        return self._plan().vc_estimator_forward(self._packed(x.device), x, x_mask, mean, ref, ref_mask, c, t)


class Diffusion(BaseModule):
    def __init__(self, n_feats, dim_unet, dim_spk, use_ref_t, beta_min, beta_max):
        super().__init__()
        self.estimator = GradLogPEstimator(dim_unet, dim_spk, use_ref_t)
        self.estimator._beta_range = (float(beta_min), float(beta_max))
        self.n_feats, self.dim_unet, self.dim_spk = n_feats, dim_unet, dim_spk
        self.use_ref_t, self.beta_min, self.beta_max = use_ref_t, beta_min, beta_max

    # ---- schedule scalars (diffusion.py:120-149)
    def get_beta(self, t):
        return self.beta_min + (self.beta_max - self.beta_min) * t

    def get_gamma(self, s, t, p=1.0, use_torch=False):
        integral = (self.beta_min + 0.5 * (self.beta_max - self.beta_min) * (t + s)) * (t - s)
        if use_torch:
            return torch.exp(-0.5 * p * integral).unsqueeze(-1).unsqueeze(-1)
        return math.exp(-0.5 * p * integral)

    def get_mu(self, s, t):
        return self.get_gamma(s, t) * (1.0 - self.get_gamma(0, s, p=2.0)) / (1.0 - self.get_gamma(0, t, p=2.0))

    def get_nu(self, s, t):
        return self.get_gamma(0, s) * (1.0 - self.get_gamma(s, t, p=2.0)) / (1.0 - self.get_gamma(0, t, p=2.0))

    def get_sigma(self, s, t):
        a, b = 1.0 - self.get_gamma(0, s, p=2.0), 1.0 - self.get_gamma(s, t, p=2.0)
        return math.sqrt(a * b / (1.0 - self.get_gamma(0, t, p=2.0)))

    def compute_diffused_mean(self, x0, mask, mean, t, use_torch=False):
        w = self.get_gamma(0, t, use_torch=use_torch)
        return (x0 * w + mean * (1.0 - w)) * mask

    def forward_diffusion(self, x0, mask, mean, t):
        xt_mean = self.compute_diffused_mean(x0, mask, mean, t, use_torch=True)
        variance = 1.0 - self.get_gamma(0, t, p=2.0, use_torch=True)
        z = torch.randn(x0.shape, dtype=x0.dtype, device=x0.device, requires_grad=False)
        return (xt_mean + z * torch.sqrt(variance)) * mask, z * mask

    @torch.no_grad()
    def reverse_diffusion(self, z, mask, mean, ref, ref_mask, mean_ref, c, n_timesteps, mode):
        """diffusion.py:164-196 on the HIP kernels (one C-ABI call for the whole loop).  For 'em' / 'ml' the per-step
        N(0,1) draws stay `torch.randn_like(z)` calls on z's device, in the reference's order."""
        if not z.is_cuda:
            raise RuntimeError("Diffusion.reverse_diffusion runs on the MI355X HIP kernels only; got a %s tensor "
                               "(there is no CPU fallback)" % z.device)
        est = self.estimator
        est._beta_range = (float(self.beta_min), float(self.beta_max))
        noise = None
        if mode != "pf":
            # the reference's draws, in its order (one randn_like(z) per step), handed over in bounded chunks
            def noise(k):
                return torch.stack([torch.randn_like(z, device=z.device) for _ in range(k)])
        return est._plan().vc_reverse_diffusion(est._packed(z.device), z, mask, mean, ref, ref_mask, mean_ref, c,
                                                n_timesteps, mode, noise)

    @torch.no_grad()
    def forward(self, z, mask, mean, ref, ref_mask, mean_ref, c, n_timesteps, mode):
        if mode not in ["pf", "em", "ml"]:
            print("Inference mode must be one of [pf, em, ml]!")
            return z
        return self.reverse_diffusion(z, mask, mean, ref, ref_mask, mean_ref, c, n_timesteps, mode)

    def loss_t(self, x0, mask, mean, x_ref, mean_ref, c, t):
        xt, z = self.forward_diffusion(x0, mask, mean, t)
        xt_ref = torch.stack([self.compute_diffused_mean(x_ref, mask, mean_ref, t, use_torch=True)], 1)
        z_est = self.estimator(xt, mask, mean, xt_ref, mask, c, t)
        z_est = z_est * torch.sqrt(1.0 - self.get_gamma(0, t, p=2.0, use_torch=True))
        return torch.sum((z_est + z) ** 2) / (torch.sum(mask) * self.n_feats)

    def compute_loss(self, x0, mask, mean, x_ref, mean_ref, c, offset=1e-5):
        t = torch.rand(x0.shape[0], dtype=x0.dtype, device=x0.device, requires_grad=False)
        return self.loss_t(x0, mask, mean, x_ref, mean_ref, c, torch.clamp(t, offset, 1.0 - offset))
