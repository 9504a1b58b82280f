#!/bin/bash
# round 5, visit F: CU-partitioned sub-batch streams (hipExtStreamCreateWithCUMask) and per-layer precision choice, alternating A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline "$@" > gpurun_out/r05f_bench_$n.json 2> gpurun_out/r05f_err_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05f_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8ws_s0_$rep --precision f16f8 --conv-ws 1 --streams 0
GTTS_F8_MIN_COUT=128 run f8ws128_s0_$rep --precision f16f8 --conv-ws 1 --streams 0
GTTS_CU_SPLIT=128:128 GTTS_WS_CUS=128 run f8ws_split_$rep --precision f16f8 --conv-ws 1 --streams 2
GTTS_F8_MIN_COUT=128 GTTS_CU_SPLIT=128:128 GTTS_WS_CUS=128 run f8ws128_split_$rep --precision f16f8 --conv-ws 1 --streams 2
GTTS_CU_SPLIT=128:128 run x3_split2_$rep --precision bf16x3 --streams 2
GTTS_CU_SPLIT=88:84:84 run x3_split3_$rep --precision bf16x3 --streams 3
GTTS_F8_MIN_COUT=128 GTTS_CU_SPLIT=88:84:84 GTTS_WS_CUS=84 run f8ws128_split3_$rep --precision f16f8 --conv-ws 1 --streams 3
done
tail -3 gpurun_out/r05f_err_f8ws_split_1.txt
