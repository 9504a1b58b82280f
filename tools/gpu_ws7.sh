#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for spec in "lag1 libgradtts_gfx950.so 0" "A32lag0 libgtts_wsA.so 0" "B22lag0 libgtts_wsB.so 0" "C22lag2 libgtts_wsC.so 0" "D33lag1 libgtts_wsD.so 0"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --per-op --streams $3 > gpurun_out/ws7_$1.json 2> gpurun_out/ws7_$1.txt
  echo "== $1 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ws7_$1.json'));print(d['value'], d['config'].get('ms_per_unet_call'), (d.get('roofline') or {}).get('avg_us'))")"
  grep -E "conv3x3_ws" gpurun_out/ws7_$1.txt | tail -4
done
