#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in t64_0 t64_1 t64_2 t64_3 t64_4 t128_0; do echo "--- trace $v"; GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so timeout 200 python tools/trace_ws.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d_trace_$v.txt; done
for s in 0 2 3 4; do
  timeout 300 python bench.py --workload gradtts-multispk --precision bf16-store --timesteps 100 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $s > gpurun_out/r3d_c3_s$s.json 2>/dev/null
  echo "config3 streams $s: $(python -c "import json;d=json.load(open('gpurun_out/r3d_c3_s$s.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
for s in 0 2 3; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $s > gpurun_out/r3d_h_s$s.json 2>/dev/null
  echo "headline streams $s: $(python -c "import json;d=json.load(open('gpurun_out/r3d_h_s$s.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
