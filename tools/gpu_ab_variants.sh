#!/bin/bash
# Same-box alternating A/B of library variants on the headline bench, plus the batch-size sweep that picks the drop-in module's plan.
# Variant libraries:  bash tools/build_variants.sh ws64off "-DGTTS_F8_WS64=0" npw4 "-DGTTS_WS64_NPW=4"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05j_bench_$n.json 2> gpurun_out/r05j_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05j_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"; }
for rep in 1 2 3; do
run bf16x3_$rep --precision bf16x3
run f8_$rep --precision f16f8
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_ws64off.so run f8_ws64off_$rep --precision f16f8
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_npw4.so run f8_npw4_$rep --precision f16f8
done
run b1_x3 --precision bf16x3 --batch 1
run b1_f8 --precision f16f8 --batch 1
run b4_x3 --precision bf16x3 --batch 4
run b4_f8 --precision f16f8 --batch 4
run b4_f8_mfma3 --precision f16f8 --batch 4 --conv-ws 0 --streams 3
run b8_x3 --precision bf16x3 --batch 8
run b8_f8 --precision f16f8 --batch 8
run b8_f8_mfma3 --precision f16f8 --batch 8 --conv-ws 0 --streams 3
