"""Drop-in mirror of Grad-TTS/model/diffusion.py for MI355X.

Same class names, constructor signatures, parameter names and shapes, so reference
checkpoints load with strict=True and the reference's inference.py / train.py drive these classes unchanged.
What differs is where the arithmetic runs:

  * sampling (torch.no_grad; GradLogPEstimator2d.forward, Diffusion.forward / reverse_diffusion):
    hand-written HIP kernels for gfx950 behind the C ABI of libgradtts_gfx950.so.  No PyTorch compute ops,
    no CPU fallback -- a missing library or a CPU tensor raises RuntimeError.
  * training (autograd enabled; Diffusion.compute_loss / loss_t): the differentiable composition in
    _train_ops.py (stock PyTorch-ROCm ops) over the very same nn.Parameters.

Reference lines are cited per symbol.
"""
import math

import torch

from . import _train_ops
from ._backend import backend
from .base import BaseModule


class Mish(BaseModule):
    """diffusion.py:16-18 (parameter-free; kept so Sequential indices match the reference's state_dict)."""

    def forward(self, x):
        return x * torch.tanh(torch.nn.functional.softplus(x))


class Upsample(BaseModule):
    """diffusion.py:21-27 -- ConvTranspose2d(dim, dim, 4, 2, 1); weight layout [Cin, Cout, 4, 4]."""

    def __init__(self, dim):
        super().__init__()
        self.conv = torch.nn.ConvTranspose2d(dim, dim, kernel_size=4, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Downsample(BaseModule):
    """diffusion.py:30-36 -- Conv2d(dim, dim, 3, stride 2, pad 1)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = torch.nn.Conv2d(dim, dim, kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Rezero(BaseModule):
    """diffusion.py:39-46 -- fn(x) * g with g initialised to 0."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn
        self.g = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return self.fn(x) * self.g


class Block(BaseModule):
    """diffusion.py:49-58 -- Conv3x3 -> GroupNorm(groups) -> Mish on x*mask, output *mask."""

    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        layers = [torch.nn.Conv2d(dim, dim_out, kernel_size=3, padding=1), torch.nn.GroupNorm(groups, dim_out), Mish()]
        self.block = torch.nn.Sequential(*layers)

    def forward(self, x, mask):
        return _train_ops._conv_gn_mish(self, x, mask)


class ResnetBlock(BaseModule):
    """diffusion.py:61-79."""

    def __init__(self, dim, dim_out, time_emb_dim, groups=8):
        super().__init__()
        self.mlp = torch.nn.Sequential(Mish(), torch.nn.Linear(time_emb_dim, dim_out))
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = torch.nn.Conv2d(dim, dim_out, kernel_size=1) if dim != dim_out else torch.nn.Identity()

    def forward(self, x, mask, time_emb):
        return _train_ops.resnet(self, x, mask, time_emb)


class LinearAttention(BaseModule):
    """diffusion.py:82-100 -- heads=4, dim_head=32; softmax over all h*w positions of k."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads = heads
        hidden = heads * dim_head
        self.to_qkv = torch.nn.Conv2d(dim, 3 * hidden, kernel_size=1, bias=False)
        self.to_out = torch.nn.Conv2d(hidden, dim, kernel_size=1)

    def forward(self, x):
        b, c, h, w = x.shape
        qkv = self.to_qkv(x).view(b, 3, self.heads, -1, h * w)
        q, k, v = qkv.unbind(1)
        ctx = torch.matmul(torch.softmax(k, dim=-1), v.transpose(-1, -2))
        out = torch.matmul(ctx.transpose(-1, -2), q).reshape(b, -1, h, w)
        return self.to_out(out)


class Residual(BaseModule):
    """diffusion.py:103-110."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, *args, **kwargs):
        return self.fn(x, *args, **kwargs) + x


class SinusoidalPosEmb(BaseModule):
    """diffusion.py:113-125."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x, scale=1000):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=x.device).float() * -(math.log(10000) / (half - 1)))
        arg = scale * x[:, None] * freq[None, :]
        return torch.cat((arg.sin(), arg.cos()), dim=-1)


def _hip_sampling(*tensors):
    """True when the call is a sampling call (no autograd graph is being recorded)."""
    return not torch.is_grad_enabled()


class GradLogPEstimator2d(BaseModule):
    """diffusion.py:128-216 -- the score U-Net.  forward() under torch.no_grad() runs on the HIP kernels."""

    def __init__(self, dim, dim_mults=(1, 2, 4), groups=8, n_spks=None, spk_emb_dim=64, n_feats=80, pe_scale=1000):
        super().__init__()
        self.dim = dim
        self.dim_mults = dim_mults
        self.groups = groups
        self.n_spks = 1 if n_spks is None else n_spks
        self.spk_emb_dim = spk_emb_dim
        self.n_feats = n_feats
        self.pe_scale = pe_scale

        if self.n_spks > 1:
            self.spk_mlp = torch.nn.Sequential(torch.nn.Linear(spk_emb_dim, 4 * spk_emb_dim), Mish(),
                                               torch.nn.Linear(4 * spk_emb_dim, n_feats))
        self.time_pos_emb = SinusoidalPosEmb(dim)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(dim, 4 * dim), Mish(), torch.nn.Linear(4 * dim, dim))

        widths = [2 + (1 if self.n_spks > 1 else 0)] + [dim * m for m in dim_mults]
        stages = list(zip(widths[:-1], widths[1:]))
        # `ups` is created before the mid blocks on purpose: it fixes the reference's registration order
        self.downs = torch.nn.ModuleList()
        self.ups = torch.nn.ModuleList()
        for i, (cin, cout) in enumerate(stages):
            last = i == len(stages) - 1
            self.downs.append(torch.nn.ModuleList([
                ResnetBlock(cin, cout, time_emb_dim=dim), ResnetBlock(cout, cout, time_emb_dim=dim),
                Residual(Rezero(LinearAttention(cout))), torch.nn.Identity() if last else Downsample(cout)]))
        mid = widths[-1]
        self.mid_block1 = ResnetBlock(mid, mid, time_emb_dim=dim)
        self.mid_attn = Residual(Rezero(LinearAttention(mid)))
        self.mid_block2 = ResnetBlock(mid, mid, time_emb_dim=dim)
        for cin, cout in reversed(stages[1:]):
            self.ups.append(torch.nn.ModuleList([
                ResnetBlock(2 * cout, cin, time_emb_dim=dim), ResnetBlock(cin, cin, time_emb_dim=dim),
                Residual(Rezero(LinearAttention(cin))), Upsample(cin)]))
        self.final_block = Block(dim, dim)
        self.final_conv = torch.nn.Conv2d(dim, 1, kernel_size=1)

        # HIP-side state (not parameters, not in the state_dict)
        self._beta_range = (0.05, 20.0)
        self._precision = None          # None -> f16f8 (fp32-grade: fp16 hi*hi + fp8 cross terms on the wide 3x3 convolutions, bf16x3 elsewhere)
        self._hip_plan = None
        self._hip_plan_key = None
        self._hip_blob = None
        self._hip_key = None
        # load_state_dict() replaces parameter contents in place (copy_ bumps _version, so the key below would catch
        # it anyway); the hook makes the repack explicit and independent of that detail
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    # ---- HIP plumbing -------------------------------------------------------------------------------
    def set_precision(self, precision):
        """'f16f8' (default; fp32-grade: ~5e-5 of max|ref| per estimator call, 4e-4 max-abs on mel-scale data after 50 Euler steps),
        'bf16x3' (fp32-grade, ~2e-5 per call, every contraction as three bf16 MFMA passes), 'bf16' (single bf16 MFMA, fp32
        activations) or 'bf16_store' (BASELINE config 3 as written: bf16 MFMA and bf16 activation storage).

        Batch-size buckets (f16f8 only).  A Plan's results never depend on how utterances are batched (bit-identical).  This MODULE
        however picks the plan by batch size (_variant): B = 1 and B >= 7 run the persistent kernel with the 64-channel layers in
        f16 + fp8, B = 2...6 the uniform-wave kernel with those layers in bf16x3 (faster there).  The same utterance sampled alone
        and inside a batch of four therefore differs by fp32-grade rounding (<= 1e-4 of max|x| after 50 steps, tested), not by
        zero.  Call set_precision('bf16x3'), or pin one bucket with pin_variant('main'), where bit-equality across batch sizes
        matters more than the B = 2...6 speed."""
        be = backend()
        self._precision = {"bf16x3": be.PREC_BF16X3, "bf16": be.PREC_BF16, "bf16_store": be.PREC_BF16_STORE, "f16f8": be.PREC_F16F8}[precision]
        self._hip_plan = None
        self.invalidate_packed()

    def invalidate_packed(self):
        """Drop the packed-weight blob; the next sampling call re-packs from the current parameters.

        The automatic check below keys on (data_ptr, Tensor._version) of every parameter, which sees optimizer steps,
        load_state_dict, .to() and ordinary in-place ops -- but NOT writes made through `p.data` (`p.data.copy_()`,
        `p.data.mul_()`, as EMA weight swaps do): those do not bump `_version`.  Call this after such an edit."""
        self._hip_blob = None
        self._hip_key = None
        backend().clear_packed_cache()          # the training kernels' packed copies (keyed on version + generation)

    def _variant(self, batch):
        """Which plan a call of `batch` utterances takes.  The default precision (f16f8) runs the Block convolutions on the persistent
        kernel, unsplit -- best at B = 1 and B >= 8; in between the uniform-wave kernel on three sub-batch streams is faster (measured
        on one box, ms per U-Net call at B = 4: 2.45 against 2.62; B = 8: 3.72 against 3.68; B = 1: 1.48 against 1.41)."""
        be = backend()
        prec = be.PREC_F16F8 if self._precision is None else self._precision
        if getattr(self, "_pinned_variant", None) is not None:
            return self._pinned_variant
        return "mid" if (prec == be.PREC_F16F8 and batch is not None and 2 <= int(batch) <= 6) else "main"

    def pin_variant(self, variant=None):
        """'main' / 'mid': always take that plan whatever the batch size (results then bit-identical across batch sizes); None: by batch."""
        if variant not in (None, "main", "mid"):
            raise ValueError("variant must be None, 'main' or 'mid'")
        self._pinned_variant = variant

    def _plan(self, batch=None):
        if tuple(self.dim_mults) != (1, 2, 4) or self.groups != 8:
            raise RuntimeError("the HIP path supports dim_mults=(1,2,4), groups=8 (the reference's configuration)")
        be = backend()
        prec = be.PREC_F16F8 if self._precision is None else self._precision
        var = self._variant(batch)
        # everything the plan captures at creation is part of the key: changing beta_min / beta_max / pe_scale on the
        # module after the first sample rebuilds the plan (the ODE sampler takes beta from the plan's cfg)
        key = (prec, float(self._beta_range[0]), float(self._beta_range[1]), float(self.pe_scale))
        if self._hip_plan is None or self._hip_plan_key != key:
            self._hip_plan = {}
            self._hip_plan_key = key
            self.invalidate_packed()
        if var not in self._hip_plan:
            kw = dict(conv_ws=False, streams=3) if var == "mid" else {}
            self._hip_plan[var] = be.Plan(dim=self.dim, n_feats=self.n_feats, n_spks=self.n_spks,
                                          spk_emb_dim=self.spk_emb_dim, groups=self.groups, pe_scale=float(self.pe_scale),
                                          beta_min=key[1], beta_max=key[2], precision=prec, **kw)
        return self._hip_plan[var]

    def _packed(self, device, batch=None):
        """Packed weights, re-packed whenever a parameter changed (optimizer step, load_state_dict, .to()); see
        invalidate_packed() for the one case the check cannot see.  (One blob per plan variant: the 64-channel layers are packed in
        the f16 + fp8 format only for the persistent kernel.)"""
        plan = self._plan(batch)
        var = self._variant(batch)
        params = list(self.named_parameters())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for _, p in params)
        if self._hip_blob is None or self._hip_key != key:
            self._hip_blob = {}
            self._hip_key = key
        if var not in self._hip_blob:
            try:
                self._hip_blob[var] = plan.pack({n: p for n, p in params}, device)
            except backend().RangeError as e:
                # the default precision has a weight range (|w| < 63.97 on the 3x3 Block convolutions, include/gradtts_abi.h); the
                # reference's fp32 path has none.  An out-of-range checkpoint is never sampled at a lower grade silently: the
                # module switches itself to bf16x3 (fp32-grade, no range limits, ~8 % slower) and says so.  An explicit
                # set_precision('f16f8') is a request, not a default: then the error is the caller's to see.
                if self._precision is not None:
                    raise
                import warnings
                warnings.warn("GradLogPEstimator2d: %s -- sampling with precision 'bf16x3' instead" % e, RuntimeWarning, stacklevel=3)
                self._precision = backend().PREC_BF16X3
                self._hip_plan = None
                self.invalidate_packed()
                return self._packed(device, batch)
        return self._hip_blob[var]

    def range_status(self):
        """(events, max |x|) of the last sampling call's activation range record (see Plan.range_status): non-zero events mean
        some activations were beyond the f16f8 cross-term range (|x| >= 1024) and were carried at fp16 grade.  Synchronises."""
        if not self._hip_plan:
            return 0, 0.0
        ev, mx = 0, 0.0
        for plan in self._hip_plan.values():
            e, m_ = plan.range_status()
            ev, mx = ev + e, max(mx, m_)
        return ev, mx

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, mask, mu, t, spk=None):
        if not _hip_sampling():
            return _train_ops.estimator(self, x, mask, mu, t, spk)
        if not x.is_cuda:
            raise RuntimeError("GradLogPEstimator2d sampling runs on the MI355X HIP kernels only; got a %s tensor "
                               "(there is no CPU fallback)" % x.device)
        if self.n_spks > 1 and spk is None:
            raise RuntimeError("multi-speaker estimator needs spk")
        blob = self._packed(x.device, x.shape[0])       # (first: an out-of-range checkpoint switches the module's precision here)
        plan = self._plan(x.shape[0])
        return plan.estimator_forward(blob, x, mask, mu, t, spk if self.n_spks > 1 else None)


def get_noise(t, beta_init, beta_term, cumulative=False):
    """diffusion.py:219-224."""
    if cumulative:
        return beta_init * t + 0.5 * (beta_term - beta_init) * (t ** 2)
    return beta_init + (beta_term - beta_init) * t


class Diffusion(BaseModule):
    """diffusion.py:227-294."""

    def __init__(self, n_feats, dim, n_spks=1, spk_emb_dim=64, beta_min=0.05, beta_max=20, pe_scale=1000):
        super().__init__()
        self.n_feats = n_feats
        self.dim = dim
        self.n_spks = n_spks
        self.spk_emb_dim = spk_emb_dim
        self.beta_min = beta_min
        self.beta_max = beta_max
        self.pe_scale = pe_scale
        self.estimator = GradLogPEstimator2d(dim, n_spks=n_spks, spk_emb_dim=spk_emb_dim, n_feats=n_feats,
                                             pe_scale=pe_scale)
        self.estimator._beta_range = (float(beta_min), float(beta_max))

    def forward_diffusion(self, x0, mask, mu, t):
        """diffusion.py:244-252.  HIP tensors: one fused kernel after the (reference-ordered) N(0,1) draw."""
        z = torch.randn(x0.shape, dtype=x0.dtype, device=x0.device, requires_grad=False)
        if x0.is_cuda and not x0.requires_grad and not mu.requires_grad and not _train_ops.FORCE_TORCH:
            return backend().diffusion_noising(x0, mu, z, mask, t, self.beta_min, self.beta_max)
        time = t[:, None, None]
        cum = get_noise(time, self.beta_min, self.beta_max, cumulative=True)
        decay = torch.exp(-0.5 * cum)
        mean = x0 * decay + mu * (1.0 - decay)
        xt = mean + z * torch.sqrt(1.0 - torch.exp(-cum))
        return xt * mask, z * mask

    @torch.no_grad()
    def reverse_diffusion(self, z, mask, mu, n_timesteps, stoc=False, spk=None):
        """diffusion.py:254-275 -- N Euler (ODE) / Euler-Maruyama (stoc) steps, all on the HIP kernels.

        ODE: one C-ABI call runs the whole loop (gtts_reverse_diffusion).  SDE: the per-step N(0,1) draw stays a
        torch.randn call of z's shape on z's device, exactly like the reference (same RNG stream consumption)."""
        if not z.is_cuda:
            raise RuntimeError("Diffusion.reverse_diffusion runs on the MI355X HIP kernels only; got a %s tensor "
                               "(there is no CPU fallback)" % z.device)
        est = self.estimator
        est._beta_range = (float(self.beta_min), float(self.beta_max))
        blob = est._packed(z.device, z.shape[0])         # (first: an out-of-range checkpoint switches the module's precision here)
        plan = est._plan(z.shape[0])
        spk_in = spk if est.n_spks > 1 else None
        if not stoc:
            return plan.reverse_diffusion(blob, z, mask, mu, n_timesteps, spk_in)
        import numpy as np
        be = backend()
        h = 1.0 / n_timesteps
        xt = (z * mask).float().contiguous()
        for i in range(n_timesteps):
            t = (1.0 - (i + 0.5) * h) * torch.ones(z.shape[0], dtype=z.dtype, device=z.device)
            # get_noise on the host in fp32, rounding after every op like the reference's tensor math (no sync)
            t32 = np.float32(1.0 - (i + 0.5) * h)
            beta_t = float(np.float32(self.beta_min) + np.float32(self.beta_max - self.beta_min) * t32)
            eps = plan.estimator_forward(blob, xt, mask, mu, t, spk_in)
            noise = torch.randn(z.shape, dtype=z.dtype, device=z.device, requires_grad=False)
            be.euler_step(xt, mu, eps, mask, beta_t, h, noise)
        return xt

    @torch.no_grad()
    def forward(self, z, mask, mu, n_timesteps, stoc=False, spk=None):
        """diffusion.py:277-279."""
        return self.reverse_diffusion(z, mask, mu, n_timesteps, stoc, spk)

    def loss_t(self, x0, mask, mu, t, spk=None):
        """diffusion.py:281-288."""
        xt, z = self.forward_diffusion(x0, mask, mu, t)
        est = self.estimator(xt, mask, mu, t, spk)
        if est.is_cuda and not _train_ops.FORCE_TORCH:
            # fused loss head: squared error reduction and d loss / d est in one pass (csrc/train.hip); the normaliser stays a
            # device scalar (no host synchronisation per training step)
            inv_denom = 1.0 / (torch.sum(mask) * self.n_feats)
            loss = _train_ops.ScoreLoss.apply(est.contiguous(), z, t, float(self.beta_min), float(self.beta_max), inv_denom)
            return loss, xt
        cum = get_noise(t[:, None, None], self.beta_min, self.beta_max, cumulative=True)
        eps = est * torch.sqrt(1.0 - torch.exp(-cum))
        loss = torch.sum((eps + z) ** 2) / (torch.sum(mask) * self.n_feats)
        return loss, xt

    def compute_loss(self, x0, mask, mu, spk=None, offset=1e-5):
        """diffusion.py:290-294."""
        t = torch.rand(x0.shape[0], dtype=x0.dtype, device=x0.device, requires_grad=False)
        return self.loss_t(x0, mask, mu, torch.clamp(t, offset, 1.0 - offset), spk)
