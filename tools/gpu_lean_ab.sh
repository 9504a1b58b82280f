#!/bin/bash
# Does code the default plan never launches cost time?  Product library vs a build without the single-pass bf16 / bf16-storage
# convolution instances (bash tools/build_variants.sh lean "-DGTTS_LEAN"), headline bench in alternating repeats on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline "$@" > gpurun_out/lean_$n.json 2> gpurun_out/lean_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/lean_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"; }
for rep in 1 2 3; do
run full_$rep
GTTS_LIB=$PWD/speech-backbones_amd/libgtts_lean.so run lean_$rep
done
