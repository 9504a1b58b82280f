#!/bin/bash
# Timing ablations of the f16 + fp8 persistent kernel (results of the ablated builds are WRONG by design).  Variant libraries:
#   bash tools/build_variants.sh wstrace "-DGTTS_DIAG -DGTTS_WS_TRACE=1" wsx1 "-DGTTS_DIAG -DGTTS_WS_TRACE=1 -DGTTS_WS_EXP=1" ... (EXP 1, 2, 3, 5)
# Output of round 5: profiles/r05_ws_f16f8_ablations.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in wstrace wsx1 wsx2 wsx3 wsx4 wsx5; do
  export GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so
  TRACE_PREC=f16f8 timeout 120 python tools/trace_ws.py > gpurun_out/r05d_trace_$v.txt 2>&1
  timeout 200 python bench.py --steps 1 --warmup 1 --timesteps 4 --no-cpu-baseline --no-extras --precision f16f8 --conv-ws 1 --streams 0 --per-op > gpurun_out/r05d_bench_$v.json 2> gpurun_out/r05d_perop_$v.txt
  echo "== $v"; grep -E "chunk loops|staging  |image wait|slot wait" gpurun_out/r05d_trace_$v.txt | cut -c1-70
  grep -E "^(downs.1.1.b1|downs.1.1.b2|downs.2.1.b2|ups.0.0.b1)" gpurun_out/r05d_perop_$v.txt | cut -c1-140
done
