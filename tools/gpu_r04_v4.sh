#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 500 python -m pytest tests/test_gpu_parity_full.py -m gpu -q -x -p no:cacheprovider -k "bf16 or config3 or 247" 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_b.json'))
print(d['value'], d['config']['ms_per_unet_call'])
e = d['extras']
print('cfg3', e['config3_bf16_store'].get('ms_per_unet_call'), e['config3_bf16_store'].get('error'))
print('train', e['train_step'])
print('b1', e['batch1'])
for k, v in e.items():
    if 'error' in v: print(k, v)
PY
grep -A30 "config 3" gpurun_out/r04_bench_b.txt | grep -E "tail_identity|final_euler"
