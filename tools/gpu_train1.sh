#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x -p no:cacheprovider > gpurun_out/tr1_tests.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/tr1_tests.txt
timeout 300 python tools/train_prof.py 2>&1 | grep step
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o prof -- python $ROOT/tools/train_prof.py > /tmp/prof_tr.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_tr -name "*kernel_stats*.csv"); do cp $f gpurun_out/tr1_kernel_stats.csv; done
python tools/stats_summary.py gpurun_out/tr1_kernel_stats.csv 45
