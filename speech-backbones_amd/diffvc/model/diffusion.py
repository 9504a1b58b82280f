"""Drop-in mirror of DiffVC/model/diffusion.py for MI355X: `GradLogPEstimator` (speaker-conditional score U-Net,
diffusion.py:17-106) and `Diffusion` with the pf / em / ml samplers (diffusion.py:109-222).  Same constructor
signatures, parameter names and shapes (117 794 599 decoder parameters at dim_unet=256).

Sampling (torch.no_grad) runs on the HIP kernels through the C ABI (gtts_vc_*); there is no CPU fallback.  Training
(loss_t / compute_loss, autograd) composes stock PyTorch-ROCm ops over the same parameters.
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

    def _forward_torch(self, x, x_mask, mean, ref, ref_mask, c, t):
        """Autograd composition (training only), diffusion.py:61-106."""
        condition = self.time_pos_emb(t)
        temb = self.mlp(condition)
        v = torch.stack([mean, x], 1)
        m0 = x_mask.unsqueeze(1)
        if self.use_ref_t:
            condition = torch.cat([condition, self.ref_block(ref, ref_mask.unsqueeze(1), temb)], 1)
        condition = self.cond_block(torch.cat([condition, c], 1))
        v = torch.cat([v, condition[:, :, None, None].expand(-1, -1, v.shape[2], v.shape[3])], 1)
        skips, pyramid = [], [m0]
        for r1, r2, att, down in self.downs:
            m = pyramid[-1]
            v = T.attention(att, T.resnet(r2, T.resnet(r1, v, m, temb), m, temb))
            skips.append(v)
            if not isinstance(down, torch.nn.Identity):
                v = down.conv(v * m)
            pyramid.append(m[..., ::2])
        pyramid.pop()
        m = pyramid[-1]
        v = T.resnet(self.mid_block2, T.attention(self.mid_attn, T.resnet(self.mid_block1, v, m, temb)), m, temb)
        for r1, r2, att, up in self.ups:
            m = pyramid.pop()
            v = torch.cat((v, skips.pop()), dim=1)
            v = T.attention(att, T.resnet(r2, T.resnet(r1, v, m, temb), m, temb))
            v = up.conv(v * m)
        v = T._conv_gn_mish(self.final_block, v, m0)
        out = torch.nn.functional.conv2d(v * m0, self.final_conv.weight, self.final_conv.bias)
        return (out * m0).squeeze(1)

    def forward(self, x, x_mask, mean, ref, ref_mask, c, t):
        if torch.is_grad_enabled():
            return self._forward_torch(x, x_mask, mean, ref, ref_mask, c, t)
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
