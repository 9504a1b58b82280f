import importlib, sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("speech-backbones_amd")
L = S._lib
dev = torch.device("cuda:0")
for (B, cin, cout, H, W) in [(16, 64, 64, 80, 172), (16, 128, 128, 40, 86), (16, 256, 256, 20, 43), (16, 512, 128, 20, 43)]:
    x = torch.randn(B, cin, H, W, device=dev); dy = torch.randn(B, cout, H, W, device=dev); m = torch.ones(B, W, device=dev)
    for _ in range(3): L.conv3x3_wgrad(x, m, dy)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): L.conv3x3_wgrad(x, m, dy)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    fl = 2.0 * B * H * W * cin * cout * 9
    print("wgrad %s: %.1f us  %.1f TF useful" % ((B, cin, cout, H, W), dt * 1e6, fl / dt / 1e12))
