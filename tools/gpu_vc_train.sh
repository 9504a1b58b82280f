#!/bin/bash
# DiffVC decoder training on the HIP kernels: gradient parity tests, the step timing of bench.py's extras.train_step_diffvc, and the
# kernel-time share of the gtts:: kernels (rocprofv3 kernel statistics of twelve steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_diffvc_training.py -m gpu -q -s -p no:cacheprovider > gpurun_out/vc_train_tests.txt 2>&1
tail -3 gpurun_out/vc_train_tests.txt; grep -E "DiffVC dim|compute_loss over|^FAILED|^ERROR|Error" gpurun_out/vc_train_tests.txt | head -20
(cd /tmp && MIOPEN_FIND_MODE=FAST timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vcp -o p -- python $ROOT/tools/vc_train_prof.py > /tmp/vcp.log 2>&1); echo "rocprof rc=$?"
f=$(find /tmp/vcp -name "*kernel_stats*.csv" | head -1); cp $f gpurun_out/vc_train_kernel_stats.csv
python tools/vc_train_prof.py --summarize gpurun_out/vc_train_kernel_stats.csv | tee gpurun_out/vc_train_share.txt
