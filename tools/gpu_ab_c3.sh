#!/bin/bash
# A/B of library variants on BASELINE config 3 (bf16 storage): args = variant names (speech-backbones_amd/libgtts_<name>.so; "product" = default).
# Two alternating repeats, default streams; then the config-3 parity tests on the product library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do for e in "$@"; do
  L=$PWD/speech-backbones_amd/libgtts_$e.so; [ $e = product ] && L=$PWD/speech-backbones_amd/libgradtts_gfx950.so
  GTTS_LIB=$L timeout 300 python bench.py --workload gradtts-multispk --precision bf16-store --timesteps 100 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/c3_$e.json 2> gpurun_out/c3_$e.txt
  echo "== $e rep $rep rc=$? $(python -c "import json;d=json.load(open('gpurun_out/c3_$e.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
done; done
for e in "$@"; do echo "-- $e"; grep -m6 "conv_mfma_kernel<0" gpurun_out/c3_$e.txt | cut -c1-120; done
timeout 600 python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -k "config3 or every_op" 2>&1 | tail -4
