#!/bin/bash
# round 5, visit A: parity of GTTS_PREC_F16F8 + alternating A/B of the headline bench (bf16x3 vs f16f8) with per-kernel tables
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f16f8.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r05a_tests.txt 2>&1
tail -5 gpurun_out/r05a_tests.txt
grep -E "f16f8|rel |FAILED|passed|failed" gpurun_out/r05a_tests.txt | head -40
for rep in 1 2; do for prec in bf16x3 f16f8; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --precision $prec > gpurun_out/r05a_bench_${prec}_$rep.json 2> gpurun_out/r05a_tables_${prec}_$rep.txt
echo "bench $prec $rep: $(python -c "import json;d=json.load(open('gpurun_out/r05a_bench_${prec}_$rep.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done; done
grep -E "conv_mfma_kernel<0" gpurun_out/r05a_tables_f16f8_1.txt | head -12
grep -E "conv_mfma_kernel<0" gpurun_out/r05a_tables_bf16x3_1.txt | head -8
