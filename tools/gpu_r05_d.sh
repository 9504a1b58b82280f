#!/bin/bash
# round 5, visit D: timing ablations of the f16 + fp8 persistent kernel (results of the ablated builds are WRONG by design)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in wstrace wsx1 wsx2 wsx3 wsx4 wsx5; do
  export GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so
  TRACE_PREC=f16f8 timeout 120 python tools/trace_ws.py > gpurun_out/r05d_trace_$v.txt 2>&1
  timeout 200 python bench.py --steps 1 --warmup 1 --timesteps 4 --no-cpu-baseline --no-extras --precision f16f8 --conv-ws 1 --streams 0 --per-op > gpurun_out/r05d_bench_$v.json 2> gpurun_out/r05d_perop_$v.txt
  echo "== $v"; grep -E "chunk loops|staging  |image wait|slot wait" gpurun_out/r05d_trace_$v.txt | cut -c1-70
  grep -E "^(downs.1.1.b1|downs.1.1.b2|downs.2.1.b2|ups.0.0.b1)" gpurun_out/r05d_perop_$v.txt | cut -c1-140
done
