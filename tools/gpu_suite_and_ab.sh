#!/bin/bash
# The whole GPU suite + smoke, then alternating A/B of consumer pipeline depths, then the driver's default bench line.
# Variant libraries:  bash tools/build_variants.sh lead160 "-DGTTS_WS_LEAD=160" w64n2 "-DGTTS_WS64_NWS=2 -DGTTS_WS64_LEAD=160"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05k_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r05k_pytest_gpu.txt; grep -E "^FAILED|^ERROR" gpurun_out/r05k_pytest_gpu.txt | head -20
run() { n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/r05k_bench_$n.json 2> gpurun_out/r05k_tables_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/r05k_bench_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"; }
for rep in 1 2; do
run x3_$rep --precision bf16x3
run f8_$rep
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_lead160.so run f8_lead160_$rep
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_w64n2.so run f8_w64n2_$rep
done
timeout 900 python bench.py > gpurun_out/r05k_bench_default.json 2> gpurun_out/r05k_bench_default_tables.txt
python -c "import json;d=json.load(open('gpurun_out/r05k_bench_default.json'));print('default:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], {k:(v.get('ms_per_unet_call') if isinstance(v,dict) else v) for k,v in d.get('extras',{}).items()})" 2>&1 | tail -3
