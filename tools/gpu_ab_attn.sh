#!/bin/bash
# A/B of library variants that must be bit-identical on the sampler output: args = variant names ("product" = default library).
# For each: sha256 of a ragged B=16 x 1024 three-step sampler call in the three precisions + DiffVC, then the per-op table rows of
# the attention kernels (single stream).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for e in "$@"; do
  L=$PWD/speech-backbones_amd/libgtts_$e.so; [ $e = product ] && L=$PWD/speech-backbones_amd/libgradtts_gfx950.so
  GTTS_LIB=$L timeout 300 python - <<PY
import importlib, torch, sys, hashlib
sys.path.insert(0, '.')
from oracle import gradtts_oracle as O
S = importlib.import_module("speech-backbones_amd")
dev = torch.device("cuda:0")
for name, prec in (("bf16x3", S.PREC_BF16X3), ("bf16", S.PREC_BF16), ("bf16_store", S.PREC_BF16_STORE)):
    plan = S.Plan(precision=prec)
    blob = plan.pack(O.make_estimator_state(seed=0), dev)
    inp = O.make_inputs(16, 1024, seed=1234, ragged=True)
    out = plan.reverse_diffusion(blob, inp["z"].to(dev), inp["mask"].to(dev), inp["mu"].to(dev), 3).cpu()
    print("$e", name, hashlib.sha256(out.numpy().tobytes()).hexdigest()[:16], float(out.abs().max()))
PY
  GTTS_STREAMS=1 GTTS_LIB=$L timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --per-op > gpurun_out/attn_$e.json 2> gpurun_out/attn_$e.txt
  echo "== $e rc=$? $(python -c "import json;d=json.load(open('gpurun_out/attn_$e.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
  grep "attn_ctx" gpurun_out/attn_$e.txt | cut -c1-130
done
