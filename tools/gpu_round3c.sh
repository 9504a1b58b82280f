#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_training.py::test_gradtts_compute_loss_multispeaker_gpu_vs_cpu" "tests/test_gpu_parity_full.py::test_graph_replay_is_bit_identical_to_eager" -m gpu -q -s -p no:cacheprovider > gpurun_out/r3c_tests.txt 2>&1; echo "pytest rc=$?"; grep -E "rel err|max\|err\||passed|failed|Error" gpurun_out/r3c_tests.txt | tail -12
timeout 300 python tools/ws_repro2.py 2>&1 | grep -v amdgpu.ids | grep -v "\[0, 0, 0\]"
timeout 600 python bench.py > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench_tables.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3c_bench.json'))
print(d['value'], d['config']['ms_per_unet_call'], d['roofline']['kernel'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline'].get('frac_algorithmic'))
for k, v in d.get('extras', {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk not in ('workload', 'roofline')} if isinstance(v, dict) else v)
    if isinstance(v, dict) and 'roofline' in v: print('   roofline:', v['roofline'].get('kernel'), v['roofline'].get('avg_us'), v['roofline'].get('frac'), v['roofline'].get('bound'))
PY
grep -A12 "per-kernel table, B=1" gpurun_out/r3c_bench_tables.txt | head -16
