#!/bin/bash
# A/B of library build variants on one GPU box visit: args are variant numbers N (speech-backbones_amd/libgtts_expN.so);
# 0 = the product library.  Single stream, per-kernel HIP-event table of the top conv kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for e in "$@"; do
  L=$PWD/speech-backbones_amd/libgtts_exp$e.so; [ $e = 0 ] && L=$PWD/speech-backbones_amd/libgradtts_gfx950.so
  GTTS_STREAMS=1 GTTS_LIB=$L timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/exp_$e.json 2> gpurun_out/exp_$e.txt
  echo "== exp$e rc=$? $(python -c "import json;d=json.load(open('gpurun_out/exp_$e.json'));print(d['value'], d['config']['ms_per_unet_call'])")"; head -6 gpurun_out/exp_$e.txt | tail -4 | cut -c1-100
done
