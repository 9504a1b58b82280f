#!/bin/bash
# full GPU suite + default bench (driver form) on one box visit
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r3f_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3f_smoke.txt
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3f_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3f_pytest.txt
timeout 600 python bench.py > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench_tables.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3f_bench.json'))
print(d['value'], d['config']['ms_per_unet_call'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline'].get('frac_algorithmic'), d.get('cpu_baseline'))
for k, v in d.get('extras', {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk not in ('workload', 'roofline')} if isinstance(v, dict) else v)
PY
