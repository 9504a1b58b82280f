#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python __graft_entry__.py --smoke > gpurun_out/ws10_smoke.txt 2>&1; rc=$?; echo "smoke rc=$rc"; tail -1 gpurun_out/ws10_smoke.txt
if [ $rc -ne 0 ]; then exit 1; fi
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > gpurun_out/ws10_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/ws10_pytest.txt
timeout 300 python tools/ws_repro2.py 2>&1 | grep -v amdgpu.ids | grep -v "\[0, 0, 0\]" | tee gpurun_out/ws10_repro.txt
for spec in "ws0 libgradtts_gfx950.so 0" "ws2 libgradtts_gfx950.so 2" "old3 libgtts_nows.so 3"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-op --streams $3 > gpurun_out/ws10_$1.json 2> gpurun_out/ws10_$1.txt
  echo "== $1 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ws10_$1.json'));print(d['value'], d['config'].get('ms_per_unet_call'), (d.get('roofline') or {}).get('avg_us'))")"
done
grep -E "conv3x3_ws" gpurun_out/ws10_ws0.txt | tail -4
for c in 256 64; do echo "--- trace cin=cout=$c"; GTTS_LIB=$PWD/speech-backbones_amd/libgtts_wsx${c}_0.so timeout 200 python tools/trace_ws.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ws10_trace$c.txt; done
