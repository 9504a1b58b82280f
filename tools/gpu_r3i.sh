#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o prof -- python $ROOT/tools/train_prof.py > $ROOT/gpurun_out/train_steps_r03.txt 2>&1); echo "rocprof train rc=$?"
f=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1); ls -la $f
python tools/train_step_stats.py $f 60 > gpurun_out/train_step_kernel_table_r03.txt; head -30 gpurun_out/train_step_kernel_table_r03.txt; grep step gpurun_out/train_steps_r03.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tt -o prof -- python $ROOT/tools/train_prof.py torch > /tmp/tt.log 2>&1); f=$(find /tmp/prof_tt -name "*kernel_trace.csv" | head -1)
python tools/train_step_stats.py $f 12 > gpurun_out/train_step_kernel_table_torch_r03.txt; head -4 gpurun_out/train_step_kernel_table_torch_r03.txt; grep step /tmp/tt.log
