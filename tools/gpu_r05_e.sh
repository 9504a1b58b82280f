#!/bin/bash
# round 5, visit E: the re-mapped f16 + fp8 consumer (32 channels x 10 rows per wave, double-buffered weights): parity, trace, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "not n50 and conv_ws" > gpurun_out/r05e_tests.txt 2>&1
tail -3 gpurun_out/r05e_tests.txt; grep -E "worst|FAILED|Error" gpurun_out/r05e_tests.txt | head
for v in wstrace wsx1 wsx5; do
  export GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so
  TRACE_PREC=f16f8 timeout 120 python tools/trace_ws.py > gpurun_out/r05e_trace_$v.txt 2>&1
  echo "== $v"; grep -E "chunk loops|staging  |image wait|slot wait|epilogues" gpurun_out/r05e_trace_$v.txt | cut -c1-70
done
unset GTTS_LIB
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05e_bench_$n.json 2> gpurun_out/r05e_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05e_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8ws_s0_$rep --precision f16f8 --conv-ws 1 --streams 0
run f8ws_s2_$rep --precision f16f8 --conv-ws 1
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05e_tables_f8ws_s0_1.txt | head -12
