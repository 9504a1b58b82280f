"""DiffVC decoder training steps under rocprofv3 (kernel-time share of the gtts:: kernels):
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vcp -o p -- python $ROOT/tools/vc_train_prof.py
  python $ROOT/tools/vc_train_prof.py --summarize /tmp/vcp/.../p_kernel_stats.csv"""
import csv
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    rows = list(csv.DictReader(open(sys.argv[2])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    gt = sum(float(r["TotalDurationNs"]) for r in rows if "gtts::" in r["Name"])
    print("kernel time %.1f ms over the run, %.1f %% of it in gtts:: kernels" % (tot / 1e6, 100 * gt / tot))
    for r in rows[:14]:
        print("%-90s %6s %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    sys.exit(0)

import torch  # noqa: E402

DV = importlib.import_module("speech-backbones_amd.diffvc.model.diffusion")
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = DV.Diffusion(80, 256, 128, True, 0.05, 20.0).to(dev)
g = torch.Generator().manual_seed(1)
B, T = 16, 128
x0, mean, xr, mr = (torch.randn(B, 80, T, generator=g).to(dev) for _ in range(4))
c = (torch.randn(B, 256, generator=g) * 0.3).to(dev)
mask = torch.ones(B, 1, T, device=dev)
for _ in range(40):
    dec.zero_grad(set_to_none=True)
    loss = dec.compute_loss(x0, mask, mean, xr, mr, c)
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss.detach()))
