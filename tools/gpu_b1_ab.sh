#!/bin/bash
# B = 1 A/B of library variants (alternating repeats on one box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --batch 1 "$@" > gpurun_out/b1ab_$n.json 2> gpurun_out/b1ab_$n.txt
  echo "bench $n: $(python -c "import json;d=json.load(open('gpurun_out/b1ab_$n.json'));print(d['value'], d['config'].get('ms_per_unet_call'))" 2>&1 | tail -1)"; }
for rep in 1 2 3; do
run base_$rep
for v in "$@"; do GTTS_LIB=$PWD/speech-backbones_amd/libgtts_$v.so run ${v}_$rep; done
done
