#!/bin/bash
# round 4, visit 1: measured ceilings, co-run probe, conv occupancy cap A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r04_ceilings.txt
import importlib, json, torch
pkg = importlib.import_module("speech-backbones_amd")
for rep in range(2):
    print(json.dumps(pkg._lib.measured_ceilings(torch.device("cuda:0"))))
PY
timeout 300 python tools/corun_probe.py 40 2>&1 | tee gpurun_out/r04_corun.txt
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_lds2.so timeout 300 python tools/corun_probe.py 40 2>&1 | tee gpurun_out/r04_corun_lds2.txt
for rep in 1 2; do
for spec in "prod3 libgradtts_gfx950.so 3" "lds2_3 libgtts_lds2.so 3" "lds2_2 libgtts_lds2.so 2" "lds2_4 libgtts_lds2.so 4" "prod4 libgradtts_gfx950.so 4"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $3 > gpurun_out/ab_$1_$rep.json 2>/dev/null
  echo "rep $rep $1: $(python -c "import json;d=json.load(open('gpurun_out/ab_$1_$rep.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
done
