#!/bin/bash
# same-box comparison of 64-channel-tile variants (tools/build_conv_variant.sh): per-op times of the level-0 / level-1 64-cout layers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for v in "$@"; do
  lib=libgtts_$v.so; [ "$v" = product ] && lib=libgradtts_gfx950.so
  GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --per-op > gpurun_out/v_$v.json 2> gpurun_out/v_$v.txt
  echo "== $v $(python -c "import json;d=json.load(open('gpurun_out/v_$v.json'));print(d['value'], d['config']['ms_per_unet_call'])" 2>&1 | tail -1)"
  grep -E "^downs.0.0.b2|^downs.0.1.b1|^ups.1.0.b1|^ups.1.1.b2" gpurun_out/v_$v.txt | awk '{printf "   %-20s %s\n", $1, $(NF-2)}'
done
