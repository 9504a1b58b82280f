#!/bin/bash
# A/B of runtime knobs on one GPU box visit: each argument "tag:ENV=val,ENV=val" runs bench without CPU baseline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  env $envs timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.txt
  echo "== $tag rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ab_$tag.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
  head -7 gpurun_out/ab_$tag.txt | tail -5 | cut -c1-100
done
