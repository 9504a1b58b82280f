#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TRACE_PREC=f16f8 GTTS_LIB=$PWD/speech-backbones_amd/libgtts_wstrace.so timeout 120 python tools/trace_ws.py > gpurun_out/r05i_trace64_npw4.txt 2>&1
tail -12 gpurun_out/r05i_trace64_npw4.txt
timeout 300 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -p no:cacheprovider -k "not n50_t1024 and conv_ws" 2>&1 | tail -2
run() { n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05i_bench_$n.json 2> gpurun_out/r05i_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05i_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"; }
for rep in 1 2; do
run npw4_$rep --precision f16f8
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_npw8.so run npw8_$rep --precision f16f8
done
grep -E "conv3x3_ws" gpurun_out/r05i_tables_npw4_1.txt | head -6
