"""Where the wave-specialised Upsample kernel (conv_up_ws.hip) spends its cycles: run against a -DGTTS_UP_WS=1 -DGTTS_UPW_TRACE=1
build (tools/build_conv_variant.sh upwtr "-DGTTS_UP_WS=1 -DGTTS_UPW_TRACE=1"; GTTS_LIB=.../libgtts_upwtr.so python tools/upw_trace.py);
one workgroup prints, per wave, total cycles and [0] barrier wait, [1] taps / request, [2] epilogue / staging + stores."""
import importlib, sys
import torch
sys.path.insert(0, '.')
S = importlib.import_module("speech-backbones_amd")
D = importlib.import_module("speech-backbones_amd.model.diffusion")
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = D.Diffusion(80, 64, 1, 64, 0.05, 20.0, 1000)
sd = {k[len("estimator."):]: v.detach().clone() for k, v in dec.state_dict().items()}
plan = S.Plan(precision=S.PREC_F16F8, conv_ws=True)
blob = plan.pack(sd, dev)
B, T = 16, 1024
g = torch.Generator().manual_seed(1)
z, mu = torch.randn(B, 80, T, generator=g), torch.randn(B, 80, T, generator=g)
mask = torch.ones(B, 1, T)
t = torch.full((B,), 0.5)
plan.estimator_forward(blob, z.to(dev), mask.to(dev), mu.to(dev), t.to(dev))
torch.cuda.synchronize()
