#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3j_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3j_tests.txt
timeout 300 python tools/train_prof.py 2>&1 | grep step
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -o prof -- python $ROOT/tools/train_prof.py > $ROOT/gpurun_out/train_steps_r03.txt 2>&1); echo "rocprof train rc=$?"
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python tools/train_step_stats.py $f 70 > gpurun_out/train_step_kernel_table_r03.txt; head -24 gpurun_out/train_step_kernel_table_r03.txt; grep -v "gtts::" gpurun_out/train_step_kernel_table_r03.txt | head -14
