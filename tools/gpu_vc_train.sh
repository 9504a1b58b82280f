#!/bin/bash
# DiffVC decoder training on the HIP kernels: gradient parity tests + the step timing of bench.py's extras.train_step_diffvc
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_diffvc_training.py -m gpu -q -s -p no:cacheprovider > gpurun_out/vc_train_tests.txt 2>&1
tail -3 gpurun_out/vc_train_tests.txt; grep -E "DiffVC dim|compute_loss over|^FAILED|^ERROR|Error" gpurun_out/vc_train_tests.txt | head -20
BENCH_EXTRAS=train_step_diffvc,train_step timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/vc_train_bench.json 2> gpurun_out/vc_train_bench.txt
python -c "import json;d=json.load(open('gpurun_out/vc_train_bench.json'));print(json.dumps(d['extras'], indent=1))"
