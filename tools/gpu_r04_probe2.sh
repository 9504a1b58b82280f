#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/corun_probe.py 40 2>&1 | tee gpurun_out/r04_corun2.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_a.json'))
print(d['value'], d['config']['ms_per_unet_call'])
print(json.dumps(d.get('measured_ceilings'), indent=0))
print(json.dumps(d.get('roofline'), indent=0))
print(json.dumps(d.get('unet_roofline'), indent=0))
print(json.dumps(d['extras'].get('e2e_inference'), indent=0))
for k, v in d['extras'].items():
    if 'error' in v: print(k, v)
PY
