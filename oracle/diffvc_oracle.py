"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- functional CPU restatement of the DiffVC decoder.

Follows DiffVC/model/diffusion.py:17-205 and DiffVC/model/modules.py:128-166 (RefBlock); the U-Net trunk is
byte-identical to Grad-TTS's and reuses oracle.gradtts_oracle.unet_body.
Pinned by tests/golden/vc_*.npz (outputs of the reference's own DiffVC modules) and the live test in
tests/test_oracle_vs_reference.py.
"""
import math

import torch
import torch.nn.functional as F

from . import gradtts_oracle as G


# ----------------------------------------------------------------------------- schedule scalars (host doubles)
def get_beta(t, beta_min=0.05, beta_max=20.0):
    """diffusion.py:120-122."""
    return beta_min + (beta_max - beta_min) * t


def get_gamma(s, t, p=1.0, beta_min=0.05, beta_max=20.0):
    """diffusion.py:124-131 (use_torch=False branch: Python doubles)."""
    integral = (beta_min + 0.5 * (beta_max - beta_min) * (t + s)) * (t - s)
    return math.exp(-0.5 * p * integral)


def get_mu(s, t, **kw):
    """diffusion.py:133-137."""
    a = get_gamma(s, t, **kw)
    b = 1.0 - get_gamma(0, s, p=2.0, **kw)
    c = 1.0 - get_gamma(0, t, p=2.0, **kw)
    return a * b / c


def get_nu(s, t, **kw):
    """diffusion.py:139-143."""
    a = get_gamma(0, s, **kw)
    b = 1.0 - get_gamma(s, t, p=2.0, **kw)
    c = 1.0 - get_gamma(0, t, p=2.0, **kw)
    return a * b / c


def get_sigma(s, t, **kw):
    """diffusion.py:145-149."""
    a = 1.0 - get_gamma(0, s, p=2.0, **kw)
    b = 1.0 - get_gamma(s, t, p=2.0, **kw)
    c = 1.0 - get_gamma(0, t, p=2.0, **kw)
    return math.sqrt(a * b / c)


def step_coefficients(t, h, mode, beta_min=0.05, beta_max=20.0):
    """(beta_t, kappa, omega, sigma) of one reverse step, diffusion.py:172,180-191."""
    kw = dict(beta_min=beta_min, beta_max=beta_max)
    beta_t = get_beta(t, **kw)
    if mode == "ml":
        kappa = get_gamma(0, t - h, **kw) * (1.0 - get_gamma(t - h, t, p=2.0, **kw))
        kappa /= (get_gamma(0, t, **kw) * beta_t * h)
        kappa -= 1.0
        omega = get_nu(t - h, t, **kw) / get_gamma(0, t, **kw)
        omega += get_mu(t - h, t, **kw)
        omega -= (0.5 * beta_t * h + 1.0)
        sigma = get_sigma(t - h, t, **kw)
    else:
        kappa, omega, sigma = 0.0, 0.0, math.sqrt(beta_t * h)
    return beta_t, kappa, omega, sigma


def compute_diffused_mean(x0, mask, mean, t, **kw):
    """diffusion.py:151-155 (scalar t)."""
    w = get_gamma(0, t, **kw)
    return (x0 * w + mean * (1.0 - w)) * mask


# ----------------------------------------------------------------------------- RefBlock
def _in_glu(sd, p, x):
    """Conv3x3 -> InstanceNorm2d(affine) -> GLU(dim=1)   (modules.py:140-157)."""
    y = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
    y = F.instance_norm(y, weight=sd[p + "1.weight"], bias=sd[p + "1.bias"], eps=1e-5)
    return F.glu(y, dim=1)


def ref_block(sd, p, x, mask, t_emb, taps=None):
    """RefBlock.forward, modules.py:160-166.  x [B,1,F,T_ref], mask [B,1,1,T_ref] -> [B,out_dim]."""
    y = _in_glu(sd, p + "block11.", x * mask)
    y = _in_glu(sd, p + "block12.", y * mask)
    y = y + F.linear(G.mish(t_emb), sd[p + "mlp1.1.weight"], sd[p + "mlp1.1.bias"])[:, :, None, None]
    y = _in_glu(sd, p + "block21.", y * mask)
    y = _in_glu(sd, p + "block22.", y * mask)
    y = y + F.linear(G.mish(t_emb), sd[p + "mlp2.1.weight"], sd[p + "mlp2.1.bias"])[:, :, None, None]
    y = _in_glu(sd, p + "block31.", y * mask)
    y = _in_glu(sd, p + "block32.", y * mask)
    if taps is not None:
        taps["ref.block32"] = y
    y = F.conv2d(y * mask, sd[p + "final_conv.weight"], sd[p + "final_conv.bias"])
    return (y * mask).sum((2, 3)) / (mask.sum((2, 3)) * x.shape[2])


# ----------------------------------------------------------------------------- estimator / sampler
def estimator_forward(sd, x, x_mask, mean, ref, ref_mask, c, t, taps=None):
    """GradLogPEstimator.forward, diffusion.py:61-106.

    x, mean [B,F,T]; x_mask [B,1,T]; ref [B,1,F,T_ref] (the diffused reference); ref_mask [B,1,T_ref];
    c [B,256]; t [B]."""
    dim_base = sd["mlp.0.weight"].shape[1]
    condition = G.sinusoidal_pos_emb(t, dim_base, 1000.0)
    t_emb = F.linear(G.mish(F.linear(condition, sd["mlp.0.weight"], sd["mlp.0.bias"])), sd["mlp.2.weight"],
                     sd["mlp.2.bias"])
    x0 = torch.stack([mean, x], 1)
    xm = x_mask.unsqueeze(1)
    rm = ref_mask.unsqueeze(1)
    if "ref_block.final_conv.weight" in sd:
        rfeat = ref_block(sd, "ref_block.", ref, rm, t_emb, taps)
        if taps is not None:
            taps["ref_feat"] = rfeat
        condition = torch.cat([condition, rfeat], 1)
    condition = torch.cat([condition, c], 1)
    cond = F.linear(G.mish(F.linear(condition, sd["cond_block.0.weight"], sd["cond_block.0.bias"])),
                    sd["cond_block.2.weight"], sd["cond_block.2.bias"])
    if taps is not None:
        taps["cond"] = cond
        taps["t_emb"] = t_emb
    cond_img = cond[:, :, None, None].expand(-1, -1, x0.shape[2], x0.shape[3])
    x0 = torch.cat([x0, cond_img], 1)
    if taps is not None:
        taps["x0"] = x0
    est = G.unet_body(sd, x0, xm, t_emb, taps)
    if taps is not None:
        taps["est"] = est
    return est


def reverse_diffusion(sd, z, mask, mean, ref, ref_mask, mean_ref, c, n_timesteps, mode, beta_min=0.05, beta_max=20.0,
                      noise=None):
    """Diffusion.reverse_diffusion, diffusion.py:164-196.  noise: optional pre-drawn [N,B,F,T] for 'em'/'ml'."""
    kw = dict(beta_min=beta_min, beta_max=beta_max)
    h = 1.0 / n_timesteps
    xt = z * mask
    for i in range(n_timesteps):
        t = 1.0 - i * h
        time = t * torch.ones(z.shape[0], dtype=z.dtype)
        beta_t, kappa, omega, sigma = step_coefficients(t, h, mode, **kw)
        xt_ref = torch.stack([compute_diffused_mean(ref, ref_mask, mean_ref, t, **kw)], 1)
        est = estimator_forward(sd, xt, mask, mean, xt_ref, ref_mask, c, time)
        if mode == "pf":
            dxt = 0.5 * (mean - xt - est) * (beta_t * h)
        else:
            dxt = (mean - xt) * (0.5 * beta_t * h + omega)
            dxt -= est * (1.0 + kappa) * (beta_t * h)
            eps = noise[i] if noise is not None else torch.randn_like(z)
            dxt += eps * sigma
        xt = (xt - dxt) * mask
    return xt


# ----------------------------------------------------------------------------- fixtures
def make_state(dim_base=256, dim_cond=128, use_ref_t=True, seed=0, rezero_g=0.02):
    """Random DiffVC estimator weights in the reference's state_dict layout (diffusion.py:18-59, modules.py:128-157)."""
    trunk = G.make_estimator_state(dim=dim_base, n_spks=1, seed=seed, rezero_g=rezero_g)
    g = torch.Generator().manual_seed(seed + 1000)

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    sd = {}
    sd["mlp.0.weight"], sd["mlp.0.bias"] = trunk["mlp.0.weight"], trunk["mlp.0.bias"]
    sd["mlp.2.weight"], sd["mlp.2.bias"] = trunk["mlp.2.weight"], trunk["mlp.2.bias"]
    cond_total = dim_base + 256
    if use_ref_t:
        base = dim_cond // 4
        for name, cout in (("mlp1", base), ("mlp2", 2 * base)):
            sd["ref_block.%s.1.weight" % name] = uni((cout, dim_base), dim_base)
            sd["ref_block.%s.1.bias" % name] = uni((cout,), dim_base)
        for name, cin, cout in (("block11", 1, 2 * base), ("block12", base, 2 * base), ("block21", base, 4 * base),
                                ("block22", 2 * base, 4 * base), ("block31", 2 * base, 8 * base),
                                ("block32", 4 * base, 8 * base)):
            sd["ref_block.%s.0.weight" % name] = uni((cout, cin, 3, 3), cin * 9)
            sd["ref_block.%s.0.bias" % name] = uni((cout,), cin * 9)
            sd["ref_block.%s.1.weight" % name] = 1.0 + 0.2 * (torch.rand(cout, generator=g) - 0.5)
            sd["ref_block.%s.1.bias" % name] = 0.2 * (torch.rand(cout, generator=g) - 0.5)
        sd["ref_block.final_conv.weight"] = uni((dim_cond, 4 * base, 1, 1), 4 * base)
        sd["ref_block.final_conv.bias"] = uni((dim_cond,), 4 * base)
        cond_total += dim_cond
    sd["cond_block.0.weight"] = uni((4 * dim_cond, cond_total), cond_total)
    sd["cond_block.0.bias"] = uni((4 * dim_cond,), cond_total)
    sd["cond_block.2.weight"] = uni((dim_cond, 4 * dim_cond), 4 * dim_cond)
    sd["cond_block.2.bias"] = uni((dim_cond,), 4 * dim_cond)
    # trunk with 2 + dim_cond input channels: regenerate the two first-layer tensors that depend on Cin
    cin0 = 2 + dim_cond
    for k, v in trunk.items():
        if k.startswith("mlp."):
            continue
        sd[k] = v
    sd["downs.0.0.block1.block.0.weight"] = uni((dim_base, cin0, 3, 3), cin0 * 9)
    sd["downs.0.0.res_conv.weight"] = uni((dim_base, cin0, 1, 1), cin0)
    return sd


def make_inputs(B, T, T_ref, n_feats=80, seed=7, ragged=True):
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(B, n_feats, T, generator=g)
    z = mean + torch.randn(B, n_feats, T, generator=g)
    ref = torch.randn(B, n_feats, T_ref, generator=g)
    mean_ref = torch.randn(B, n_feats, T_ref, generator=g)
    c = torch.randn(B, 256, generator=g) * 0.3
    if ragged and B > 1:
        lengths = torch.tensor([max(1, int(round(T * (1.0 - 0.25 * i / max(1, B - 1))))) for i in range(B)])
        rlen = torch.tensor([max(1, int(round(T_ref * (1.0 - 0.3 * i / max(1, B - 1))))) for i in range(B)])
    else:
        lengths = torch.full((B,), T, dtype=torch.long)
        rlen = torch.full((B,), T_ref, dtype=torch.long)
    mask = G.sequence_mask(lengths, T).unsqueeze(1).float()
    ref_mask = G.sequence_mask(rlen, T_ref).unsqueeze(1).float()
    return dict(z=z, mean=mean, mask=mask, ref=ref, ref_mask=ref_mask, mean_ref=mean_ref, c=c)
