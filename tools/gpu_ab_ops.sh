#!/bin/bash
# same-box A/B of library variants on chosen ops of the per-op table:  OPS="regex" tools/gpu_ab_ops.sh v1 v2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for r in 1 2; do
for v in "$@"; do
  lib=libgtts_$v.so; [ "$v" = product ] && lib=libgradtts_gfx950.so
  GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --per-op > gpurun_out/v_$v.json 2> gpurun_out/v_$v.txt
  echo "== $v $(python -c "import json;d=json.load(open('gpurun_out/v_$v.json'));print(d['value'], d['config']['ms_per_unet_call'], d['config']['output_finite'])" 2>&1 | tail -1)"
  grep -E "$OPS" gpurun_out/v_$v.txt
done; done
