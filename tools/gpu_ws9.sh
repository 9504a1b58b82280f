#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python __graft_entry__.py --smoke > gpurun_out/ws9_smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 gpurun_out/ws9_smoke.txt
if [ $rc -ne 0 ]; then exit 1; fi
for st in 0 2 3; do
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $st > gpurun_out/ws9_s$st.json 2> gpurun_out/ws9_s$st.txt
  echo "== streams $st rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ws9_s$st.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_nows.so timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams 3 > gpurun_out/ws9_old3.json 2>/dev/null
echo "== old3 $(python -c "import json;d=json.load(open('gpurun_out/ws9_old3.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
