#!/bin/bash
# default bench + rocprofv3 stats of the same command (both convolution kernels) + one training step's kernel table, one box visit
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench_tables.txt; echo "bench rc=$?"; cut -c1-200 gpurun_out/final_bench.json
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_f.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_f -name "*kernel_stats*.csv"); do cp $f gpurun_out/final_rocprof_kernel_stats.csv; done
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fw -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --conv-ws 1 > /tmp/prof_fw.log 2>&1); echo "rocprof ws rc=$?"
for f in $(find /tmp/prof_fw -name "*kernel_stats*.csv"); do cp $f gpurun_out/final_rocprof_kernel_stats_conv_ws.csv; done
head -2 gpurun_out/final_rocprof_kernel_stats.csv | cut -c1-150; head -2 gpurun_out/final_rocprof_kernel_stats_conv_ws.csv | cut -c1-150
bash tools/gpu_train_table.sh
