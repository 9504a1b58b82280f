#!/bin/bash
# round 5, visit B: GTTS_PREC_F16F8 on both convolution kernels -- parity, then alternating A/B of the headline bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "not n50" > gpurun_out/r05b_tests.txt 2>&1
tail -5 gpurun_out/r05b_tests.txt
grep -E "f16f8|rel |FAILED|passed|failed" gpurun_out/r05b_tests.txt | head -40
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05b_bench_$n.json 2> gpurun_out/r05b_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05b_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8ws_s2_$rep --precision f16f8 --conv-ws 1
run f8ws_s0_$rep --precision f16f8 --conv-ws 1 --streams 0
run f8ws_s3_$rep --precision f16f8 --conv-ws 1 --streams 3
run x3ws_s2_$rep --precision bf16x3 --conv-ws 1
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05b_tables_f8ws_s2_1.txt | head -12
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05b_tables_x3ws_s2_1.txt | head -8
timeout 900 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "n50" > gpurun_out/r05b_tests_n50.txt 2>&1
grep -E "f16f8|FAILED|passed|failed" gpurun_out/r05b_tests_n50.txt | head
