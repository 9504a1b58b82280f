"""Diagnostic: per-wave s_memtime sums of the wave-specialised 3x3 conv (library built with -DGTTS_DIAG -DGTTS_WS_TRACE=1)."""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = importlib.import_module("speech-backbones_amd")
B, T = int(os.environ.get("TRACE_B", "16")), 1024
dev = torch.device("cuda:0")
def _fixture_state():
    """torch.manual_seed(0) default init of the reference architecture with Rezero.g = 0.02, from the product's own module (as bench.py)."""
    D = importlib.import_module("speech-backbones_amd.model.diffusion")
    torch.manual_seed(0)
    dec = D.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
    sd = {k[len("estimator."):]: v.detach().clone() for k, v in dec.state_dict().items()}
    for k in sd:
        if k.endswith(".fn.g"):
            sd[k].fill_(0.02)
    return sd


sd = _fixture_state()
prec = {"bf16x3": S.PREC_BF16X3, "f16f8": S.PREC_F16F8}[os.environ.get("TRACE_PREC", "bf16x3")]
plan = S.Plan(n_spks=1, streams=0, conv_ws=True, precision=prec)
print("precision", os.environ.get("TRACE_PREC", "bf16x3"), "conv_ws", plan.conv_ws)
packed = plan.pack(sd, dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 80, T, generator=g).to(dev)
mu = torch.randn(B, 80, T, generator=g).to(dev)
mask = torch.ones(B, 1, T, device=dev)
t = torch.full((B,), 0.5, device=dev)
for _ in range(2):
    out = plan.estimator_forward(packed, x, mask, mu, t)
torch.cuda.synchronize()
lib = S._lib.lib()
buf = (ctypes.c_ulonglong * 8192)()
rc = lib.gtts_debug_trace_ws(buf, 8192)
a = np.array(buf[:], dtype=np.float64).reshape(64, 16, 8)
sel = a[:, 0, 4] > 0
a = a[sel]
ncw = int(os.environ.get("TRACE_NCW", "4"))      # consumer waves of the traced form (small-launch form: 1)
npw = int((a[0, ncw:, 4] > 0).sum())            # producer waves that wrote a total (4, or 8 in the 64-channel tile of the f16 + fp8 form)
print("rc", rc, "workgroups traced", a.shape[0], "(last traced launch of the call); producer waves", npw)
c, p = a[:, :ncw], a[:, ncw:ncw + npw]
items = c[:, :, 3].mean()
print("consumer: items %.0f  total %.0f cycles" % (items, c[:, :, 4].mean()))
for i, n in enumerate(["image wait", "chunk loops", "tile epilogues"]):
    print("  %-16s %10.0f  per item %8.0f  (%4.1f%%)   per wave %s" % (n, c[:, :, i].mean(), c[:, :, i].mean() / items, 100 * c[:, :, i].mean() / c[:, :, 4].mean(), " ".join("%6.0f" % (c[:, w, i].mean() / items) for w in range(ncw))))
print("producer: total %.0f cycles" % p[:, :, 4].mean())
for i, n in enumerate(["slot wait", "staging", "request setup", "finish_tile"]):
    print("  %-16s %10.0f  per item %8.0f  (%4.1f%%)" % (n, p[:, :, i].mean(), p[:, :, i].mean() / items, 100 * p[:, :, i].mean() / p[:, :, 4].mean()))
print("  (first producer wave finish_tile: %.0f)" % p[:, 0, 3].mean())

# phase groups (GTTS_WS_DEPHASE): when the first / last tile epilogue of a consumer wave began, cycles since kernel entry
grp = c[:, 0, 7] if not os.environ.get('TRACE_W64') else c[:, 0, 7] * 0
ntile = max(1.0, c[:, :, 3].mean() / float(os.environ.get("TRACE_CHUNKS", "4")))
for gval in sorted(set(grp.tolist())):
    m = grp == gval
    print("phase group %d: %d workgroups, first epilogue at %.0f, last at %.0f, epilogue cycles per tile %.0f, total %.0f" % (
        int(gval), int(m.sum()), c[m][:, :, 5].mean(), c[m][:, :, 6].mean(), c[m][:, :, 2].mean() / ntile, c[m][:, :, 4].mean()))
if os.environ.get("TRACE_W64"):
    print("64-channel tile (weights through the LDS ring): vmcnt(0) waits of the stages %.0f cycles per item" % (c[:, :, 7].mean() / items))
