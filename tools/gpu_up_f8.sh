#!/bin/bash
# f16 + fp8 Upsample: parity subset, then same-box alternating A/B against the bf16x3 Upsample build (libgtts_upbf.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -p no:cacheprovider -x -k "taps or n50_t1024_vs_oracle" 2>&1 | tail -5
for r in 1 2; do
for v in product upbf ${UP_EXTRA}; do
  lib=libgtts_$v.so; [ "$v" = product ] && lib=libgradtts_gfx950.so
  GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --per-op > gpurun_out/v_$v.json 2> gpurun_out/v_$v.txt
  echo "== $v $(python -c "import json;d=json.load(open('gpurun_out/v_$v.json'));print(d['value'], d['config']['ms_per_unet_call'], d['config']['output_finite'])" 2>&1 | tail -1)"
  grep -E "^ups.[01].3 " gpurun_out/v_$v.txt
done; done
