#!/bin/bash
# Timing ablations of the small-launch form (one consumer wave) of the f16 + fp8 persistent kernel at B = 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in ${@:-s128 s128x1 s128x3 s128x5 s256}; do
  export GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so
  TRACE_B=1 TRACE_NCW=1 TRACE_PREC=f16f8 timeout 120 python tools/trace_ws.py > gpurun_out/wss_trace_$v.txt 2>&1
  echo "== $v"; grep -E "consumer:|producer:|chunk loops|staging  |image wait|slot wait|epilogues|request setup|workgroups" gpurun_out/wss_trace_$v.txt | cut -c1-90
done
