#!/bin/bash
# Timing ablations of the 64-channel tile of the f16 + fp8 persistent kernel (ablated builds compute WRONG results by design):
#   bash tools/build_variants.sh t64 "-DGTTS_DIAG -DGTTS_WS_TRACE=1 -DGTTS_TRACE_CIN=64" t64x1 "... -DGTTS_WS_EXP=1" (EXP 1, 2, 3, 5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in t64 t64x1 t64x5 t64x3 t64x2; do
  export GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so
  TRACE_PREC=f16f8 timeout 120 python tools/trace_ws.py > gpurun_out/ws64_trace_$v.txt 2>&1
  echo "== $v"; grep -E "chunk loops|staging  |image wait|slot wait|epilogues|request setup" gpurun_out/ws64_trace_$v.txt | cut -c1-72
done
