#!/bin/bash
# same-box per-kernel comparison of library variants (single stream, HIP-event kernel table): args = variant names
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for e in "$@"; do
  L=$PWD/speech-backbones_amd/libgtts_$e.so; [ $e = product ] && L=$PWD/speech-backbones_amd/libgradtts_gfx950.so
  GTTS_STREAMS=1 GTTS_LIB=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/abk_$e.json 2> gpurun_out/abk_$e.txt
  echo "== $e $(python -c "import json;d=json.load(open('gpurun_out/abk_$e.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
  grep -v amdgpu gpurun_out/abk_$e.txt | head -14 | cut -c1-125
done
