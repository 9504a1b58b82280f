#!/bin/bash
# round 5, visit C: phase trace of the persistent kernel in both splits, parity subset, A/B of the headline bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for prec in bf16x3 f16f8; do
TRACE_PREC=$prec GTTS_LIB=$PWD/speech-backbones_amd/libgtts_wstrace.so timeout 200 python tools/trace_ws.py > gpurun_out/r05c_trace_ws_$prec.txt 2>&1
cat gpurun_out/r05c_trace_ws_$prec.txt | tail -14
done
timeout 600 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "not n50" > gpurun_out/r05c_tests.txt 2>&1
tail -2 gpurun_out/r05c_tests.txt
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05c_bench_$n.json 2> gpurun_out/r05c_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05c_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8ws_s0_$rep --precision f16f8 --conv-ws 1 --streams 0
run f8ws_s2_$rep --precision f16f8 --conv-ws 1
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05c_tables_f8ws_s0_1.txt | head -12
