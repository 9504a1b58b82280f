import importlib, sys, torch
sys.path.insert(0, '/root/repo')
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
sd = O.make_estimator_state(seed=0)
plan = S.Plan(precision=S.PREC_F16F8, conv_ws=True)
blob = plan.pack(sd, dev)
inp = O.make_inputs(16, 1024, seed=1)
t = torch.full((16,), 0.5)
out = plan.estimator_forward(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), t.to(dev))
torch.cuda.synchronize()
