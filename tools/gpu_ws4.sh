#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for c in 256 64; do for e in 0 1 2 3 4; do echo "--- trace cin=cout=$c EXP=$e"; GTTS_LIB=$PWD/speech-backbones_amd/libgtts_wsx${c}_$e.so timeout 200 python tools/trace_ws.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ws4_trace${c}_$e.txt; done; done
