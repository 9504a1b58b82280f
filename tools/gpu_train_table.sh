#!/bin/bash
# one steady-state training step as a per-kernel table (rocprofv3 kernel trace of tools/train_prof.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -o prof -- python $ROOT/tools/train_prof.py > $ROOT/gpurun_out/train_steps.txt 2>&1); echo "rocprof train rc=$?"
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python tools/train_step_stats.py $f 70 > gpurun_out/train_step_kernel_table.txt; head -12 gpurun_out/train_step_kernel_table.txt
