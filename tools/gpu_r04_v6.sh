#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fenced.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for spec in "3 0" "2 0" "0 0" "2 1" "0 1"; do set -- $spec
timeout 300 python tools/timeline_untraced.py --streams $1 --conv-ws $2 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04_timeline_untraced.txt
done
for rep in 1 2; do for lib in libgradtts_gfx950.so libgtts_c1w3.so; do
GTTS_LIB=$PWD/speech-backbones_amd/$lib timeout 300 python bench.py --workload hifigan --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
done; done
timeout 200 python -c "
import importlib, json, torch
pkg = importlib.import_module('speech-backbones_amd')
print(json.dumps(pkg._lib.measured_ceilings(torch.device('cuda:0'))))" 2>&1 | grep -v amdgpu | tee gpurun_out/r04_ceilings2.txt
