"""GPU diagnostic (not a pytest file): run the HIP estimator with keep_intermediates and compare EVERY op output
with the CPU oracle twice -- end-to-end (vs the oracle's own taps) and locally (oracle op applied to the HIP
path's own inputs), so one run isolates every faulty kernel configuration.  Writes a table to stdout.

    python tests/gpu_diag.py [B T]
"""
import importlib
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gradtts_oracle as O  # noqa: E402

S = importlib.import_module("speech-backbones_amd")


def rel(a, b):
    d = float((a - b).abs().max())
    s = float(b.abs().max())
    return d, s, d / (s + 1e-30)


def row(name, a, b, tag=""):
    if a.shape != b.shape:
        print("%-28s SHAPE MISMATCH %s vs %s" % (name, tuple(a.shape), tuple(b.shape)))
        return 1e9
    d, s, r = rel(a, b)
    nan = bool(torch.isnan(a).any())
    flag = "  <-- BAD" if (r > 2e-4 or nan) else ""
    print("%-28s %-18s ref|max| %10.4g  abs %10.3e  rel %9.2e %s%s%s" %
          (name, tuple(a.shape), s, d, r, tag, " NaN" if nan else "", flag))
    return r


def local_checks(sd, hip, mask, taps, n_spks):
    """Oracle op applied to the HIP path's own input tensors."""
    m0 = mask.unsqueeze(1)
    masks = [m0, m0[..., ::2], m0[..., ::4]]
    print("---- local (per-op) checks: oracle op on HIP inputs")

    def gn_apply(raw, p):
        return O.mish(F.group_norm(raw, 8, sd[p + "block.1.weight"], sd[p + "block.1.bias"], eps=1e-5))

    def resnet(name, xin, lvl):
        m = masks[lvl]
        p = name + "."
        exp = F.conv2d(xin * m, sd[p + "block1.block.0.weight"], sd[p + "block1.block.0.bias"], padding=1)
        row(name + ".b1.raw", hip[name + ".b1.raw"], exp, "[local]")
        b1 = hip[name + ".b1.raw"]
        tb = taps[name + ".tb"]
        h = gn_apply(b1, p + "block1.") * m + tb[:, :, None, None]
        exp = F.conv2d(h * m, sd[p + "block2.block.0.weight"], sd[p + "block2.block.0.bias"], padding=1)
        row(name + ".b2.raw", hip[name + ".b2.raw"], exp, "[local]")
        b2 = hip[name + ".b2.raw"]
        h2 = gn_apply(b2, p + "block2.") * m
        if (p + "res_conv.weight") in sd:
            res = F.conv2d(xin * m, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])
        else:
            res = xin * m
        row(name + ".out", hip[name + ".out"], h2 + res, "[local]")
        # GroupNorm scale/shift derived from the HIP raw tensor
        for blk, raw in (("b1", b1), ("b2", b2)):
            C = raw.shape[1]
            g = raw.view(raw.shape[0], 8, -1)
            mean = g.mean(-1)
            rstd = 1.0 / torch.sqrt(g.var(-1, unbiased=False) + 1e-5)
            gamma = sd[p + "block%s.block.1.weight" % blk[1]]
            beta = sd[p + "block%s.block.1.bias" % blk[1]]
            sc = gamma[None, :] * rstd.repeat_interleave(C // 8, 1)
            sh = beta[None, :] - mean.repeat_interleave(C // 8, 1) * sc
            row(name + ".%s.sc" % blk, hip[name + ".%s.sc" % blk].view(-1, C), sc, "[local]")
            row(name + ".%s.sh" % blk, hip[name + ".%s.sh" % blk].view(-1, C), sh, "[local]")
        return hip[name + ".out"]

    def attn(name, xin):
        exp = O.attn_residual(sd, name + ".", xin)
        row(name + ".out", hip[name + ".out"], exp, "[local]")
        return hip[name + ".out"]

    x = hip["x0"]
    hidden = []
    for lv in range(3):
        x = resnet("downs.%d.0" % lv, x, lv)
        x = resnet("downs.%d.1" % lv, x, lv)
        x = attn("downs.%d.2" % lv, x)
        hidden.append(x)
        if lv < 2:
            exp = F.conv2d(x * masks[lv], sd["downs.%d.3.conv.weight" % lv], sd["downs.%d.3.conv.bias" % lv], stride=2,
                           padding=1)
            row("downs.%d.3.out" % lv, hip["downs.%d.3.out" % lv], exp, "[local]")
            x = hip["downs.%d.3.out" % lv]
    x = resnet("mid_block1", x, 2)
    x = attn("mid_attn", x)
    x = resnet("mid_block2", x, 2)
    for u in range(2):
        lv = 2 - u
        x = torch.cat((x, hidden.pop()), 1)
        x = resnet("ups.%d.0" % u, x, lv)
        x = resnet("ups.%d.1" % u, x, lv)
        x = attn("ups.%d.2" % u, x)
        exp = F.conv_transpose2d(x * masks[lv], sd["ups.%d.3.conv.weight" % u], sd["ups.%d.3.conv.bias" % u], stride=2,
                                 padding=1)
        row("ups.%d.3.out" % u, hip["ups.%d.3.out" % u], exp, "[local]")
        x = hip["ups.%d.3.out" % u]
    exp = F.conv2d(x * m0, sd["final_block.block.0.weight"], sd["final_block.block.0.bias"], padding=1)
    row("final_block.raw", hip["final_block.raw"], exp, "[local]")
    fb = gn_apply(hip["final_block.raw"], "final_block.") * m0
    est = (F.conv2d(fb * m0, sd["final_conv.weight"], sd["final_conv.bias"]) * m0).squeeze(1)
    return est


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_spks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0), "| B=%d T=%d n_spks=%d" % (B, T, n_spks))
    sd = O.make_estimator_state(seed=0, n_spks=n_spks)
    inp = O.make_inputs(B, T, seed=1234, spk_dim=64 if n_spks > 1 else None)
    t = torch.linspace(0.15, 0.9, B)
    taps = {}
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["mu"], t, inp.get("spk"), taps=taps)

    plan = S.Plan(n_spks=n_spks, keep_intermediates=True)
    blob = plan.pack(sd, dev)
    torch.cuda.synchronize()
    print("packed weights: %.1f MB, workspace %.1f MB" % (plan.packed_bytes() / 1e6, plan.workspace_bytes(B, T) / 1e6))
    out = plan.estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev),
                                 inp["spk"].to(dev) if n_spks > 1 else None)
    torch.cuda.synchronize()
    hip = {k: v.detach().cpu().clone() for k, v in plan.tensors(B, T, dev).items()}
    out = out.cpu()

    print("---- end-to-end checks: HIP tensor vs oracle tap")
    # time-bias rows: [B][tb_stride]; compare the per-resnet slices and the raw time embedding
    stride = hip["tb"].shape[1]
    tb = hip["tb"].view(B, stride)
    off = 0
    order = ["downs.0.0", "downs.0.1", "downs.1.0", "downs.1.1", "downs.2.0", "downs.2.1", "mid_block1", "mid_block2",
             "ups.0.0", "ups.0.1", "ups.1.0", "ups.1.1"]
    for n in order:
        c = taps[n + ".tb"].shape[1]
        row(n + ".tb", tb[:, off:off + c], taps[n + ".tb"])
        off += c
    row("t_emb", tb[:, off:off + 64], taps["t_emb"])
    for name in taps:
        if name in hip and name not in ("t_emb",) and not name.endswith(".tb"):
            row(name, hip[name], taps[name])
    row("est (output)", out, ref)

    est_local = local_checks(sd, hip, inp["mask"], taps, n_spks)
    row("est (output)", out, est_local, "[local]")

    # ---- timing at the benchmark shape
    if os.environ.get("DIAG_TIMING", "1") == "1":
        Bb, Tb = 16, 1024
        p2 = S.Plan()
        blob2 = p2.pack(O.make_estimator_state(seed=0), dev)
        big = O.make_inputs(Bb, Tb, seed=1, ragged=False)
        z, mk, mu = big["z"].to(dev), big["mask"].to(dev), big["mu"].to(dev)
        for n in (1, 2):
            torch.cuda.synchronize()
            t0 = time.time()
            y = p2.reverse_diffusion(blob2, z, mk, mu, n)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print("reverse_diffusion B=%d T=%d N=%d: %.1f ms  (finite=%s)" % (Bb, Tb, n, dt * 1e3,
                                                                          bool(torch.isfinite(y).all())))


if __name__ == "__main__":
    main()
