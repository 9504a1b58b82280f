import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module("speech-backbones_amd")
L = S._lib
dev = torch.device("cuda:0")
B, cin, cout, H, W = 1, 64, 64, 4, 32
x = torch.ones(B, cin, H, W); m = torch.ones(B, W)
# weight of every dy position in db: dy one-hot sweeps folded into one call per channel: channel c has its one-hot at pixel c
dy = torch.zeros(B, cout, H, W)
for c in range(cout):
    dy.view(B, cout, -1)[0, c, c] = 1.0
dw, db = L.conv3x3_wgrad(x.to(dev), m.to(dev), dy.to(dev))
print("db for one-hot at pixel c (expect all 1):", [int(v) for v in db.cpu().tolist()])
dy = torch.zeros(B, cout, H, W)
for c in range(cout):
    dy.view(B, cout, -1)[0, c, 64 + c] = 1.0
dw, db = L.conv3x3_wgrad(x.to(dev), m.to(dev), dy.to(dev))
print("db for one-hot at pixel 64+c:", [int(v) for v in db.cpu().tolist()])
