#!/bin/bash
# The fused GroupNorm finalize with the textbook agent-scope release / acquire fences (-DGTTS_FENCED_FINALIZE=1 build,
# speech-backbones_amd/libgtts_fenced.so) against the product's write-through form: same sampler call, outputs must be bit-equal.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for lib in libgradtts_gfx950.so libgtts_fenced.so; do
  GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 300 python - <<PY
import importlib, torch, sys, hashlib
sys.path.insert(0, '.')
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
plan = S.Plan()
blob = plan.pack(O.make_estimator_state(seed=0), dev)
inp = O.make_inputs(16, 1024, seed=1234, ragged=True)
out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 3).cpu()
print("$lib", hashlib.sha256(out.numpy().tobytes()).hexdigest()[:16], float(out.abs().max()))
PY
done
