#!/bin/bash
# same-box A/B of library variants on the headline configuration (default streams): args = variant names, three alternating repeats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do for e in "$@"; do
  L=$PWD/speech-backbones_amd/libgtts_$e.so; [ $e = product ] && L=$PWD/speech-backbones_amd/libgradtts_gfx950.so
  GTTS_LIB=$L timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/ab_$e.json 2>/dev/null
  echo "== $e rep $rep $(python -c "import json;d=json.load(open('gpurun_out/ab_$e.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
done; done
