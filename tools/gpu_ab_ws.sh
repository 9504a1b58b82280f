#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
for spec in "ws2 libgradtts_gfx950.so 2" "old3 libgtts_nows.so 3" "ws3 libgradtts_gfx950.so 3" "old2 libgtts_nows.so 2"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $3 > gpurun_out/ab_$1_$rep.json 2>/dev/null
  echo "rep $rep $1: $(python -c "import json;d=json.load(open('gpurun_out/ab_$1_$rep.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
done
