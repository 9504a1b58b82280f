import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("speech-backbones_amd")
L = S._lib
dev = torch.device("cuda:0")
import torch.nn.functional as F
for (B, cin, cout, H, W) in [(2, 64, 64, 80, 64), (1, 64, 64, 5, 37), (3, 128, 64, 10, 17)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, H, W, generator=g); dy = torch.randn(B, cout, H, W, generator=g); m = torch.ones(B, W)
    xr = x.clone().requires_grad_(True); w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    y = F.conv2d(xr * m[:, None, None, :], w, None, padding=1); y.backward(dy)
    dw, db = L.conv3x3_wgrad(x.to(dev), m.to(dev), dy.to(dev))
    e = (dw.cpu() - w.grad).abs().max() / w.grad.abs().max()
    eb = (db.cpu() - dy.sum((0, 2, 3))).abs().max() / dy.sum((0, 2, 3)).abs().max()
    # per-tap error to localise
    et = [(float((dw.cpu()[:, :, ky, kx] - w.grad[:, :, ky, kx]).abs().max() / w.grad.abs().max())) for ky in range(3) for kx in range(3)]
    print((B, cin, cout, H, W), "rel err dw %.2e db %.2e" % (float(e), float(eb)), ["%.1e" % v for v in et])
