#!/bin/bash
# round 5, visit G: DiffVC in f16f8 (parity + config-4 bench), B = 1 in both precisions, headline A/B with the productised defaults
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_diffvc.py tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider -k "not n50_t1024" > gpurun_out/r05g_tests.txt 2>&1
tail -4 gpurun_out/r05g_tests.txt; grep -E "FAILED|Error" gpurun_out/r05g_tests.txt | head
run() { # name, args...
  n=$1; shift
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05g_bench_$n.json 2> gpurun_out/r05g_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05g_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"
}
for rep in 1 2; do
run bf16x3_$rep --precision bf16x3
run f8_$rep --precision f16f8
run vc_x3_$rep --workload diffvc --precision bf16x3
run vc_f8_$rep --workload diffvc --precision f16f8
run b1_x3_$rep --precision bf16x3 --batch 1
run b1_f8_$rep --precision f16f8 --batch 1
run b1_f8mfma_$rep --precision f16f8 --batch 1 --conv-ws 0
run b4_x3_$rep --precision bf16x3 --batch 4
run b4_f8_$rep --precision f16f8 --batch 4
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05g_tables_vc_f8_1.txt | head -8
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/r05g_tables_vc_x3_1.txt | head -6
