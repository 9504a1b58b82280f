#!/bin/bash
# Round-end evidence on ONE GPU box visit, so that the numbers agree with each other: smoke, the full GPU test suite, the
# default bench (driver form) and rocprofv3 --kernel-trace --stats of the same command with either convolution kernel.
# Outputs in gpurun_out/ (copy what is to be kept into profiles/ as r03_*).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 200 python __graft_entry__.py --smoke > gpurun_out/final_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/final_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_pytest.txt
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench_tables.txt; echo "bench rc=$?"; cut -c1-400 gpurun_out/final_bench.json
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_f.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_f -name "*kernel_stats*.csv"); do cp $f gpurun_out/final_rocprof_kernel_stats.csv; done
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fw -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --conv-ws 1 > /tmp/prof_fw.log 2>&1); echo "rocprof ws rc=$?"
for f in $(find /tmp/prof_fw -name "*kernel_stats*.csv"); do cp $f gpurun_out/final_rocprof_kernel_stats_conv_ws.csv; done
head -4 gpurun_out/final_rocprof_kernel_stats.csv | cut -c1-150; head -3 gpurun_out/final_rocprof_kernel_stats_conv_ws.csv | cut -c1-150
