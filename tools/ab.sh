#!/bin/bash
# A/B of tuning variants on one GPU box visit: each line "<tag> <env...>" runs bench without CPU baseline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.txt; echo "== $tag rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ab_$tag.json'));print(d['value'], d['config']['ms_per_unet_call'])")"; head -8 gpurun_out/ab_$tag.txt | tail -7; }
"$@"
