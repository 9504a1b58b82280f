#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/train_prof.py 2>&1 | grep step
bash tools/gpu_train_table.sh 2>&1 | head -8
grep -c . gpurun_out/train_step_kernel_table.txt; grep "pack" gpurun_out/train_step_kernel_table.txt
timeout 300 python - <<'PY'
import importlib, json, sys, torch
sys.path.insert(0, '.')
import bench as BN
S = importlib.import_module("speech-backbones_amd")
print(json.dumps(BN.extras.__code__.co_consts and "", indent=0))
PY
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_c.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_c.json')); print(d['value'], d['extras']['train_step'])"
