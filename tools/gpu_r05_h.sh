#!/bin/bash
# round 5, visit H: the 64-channel tile of the f16 + fp8 persistent kernel (4 consumer + 8 producer waves): parity, trace, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "not n50_t1024" > gpurun_out/r05h_tests.txt 2>&1
tail -3 gpurun_out/r05h_tests.txt; grep -E "worst|FAILED|Error" gpurun_out/r05h_tests.txt | head
TRACE_PREC=f16f8 GTTS_LIB=$PWD/speech-backbones_amd/libgtts_wstrace.so timeout 120 python tools/trace_ws.py > gpurun_out/r05h_trace64.txt 2>&1
grep -E "items|chunk loops|staging  |image wait|slot wait|epilogues" gpurun_out/r05h_trace64.txt | cut -c1-80
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05h_bench_$n.json 2> gpurun_out/r05h_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05h_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8_$rep --precision f16f8
run b1_f8_$rep --precision f16f8 --batch 1
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05h_tables_f8_1.txt | head -8
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --precision f16f8 --per-op > /dev/null 2> gpurun_out/r05h_perop_f8.txt
grep -E "\.conv " gpurun_out/r05h_perop_f8.txt | cut -c1-130 | head -30
