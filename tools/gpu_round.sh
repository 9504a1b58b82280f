#!/bin/bash
# One GPU-box visit: parity tests, bench (with per-kernel HIP-event table), rocprofv3 kernel trace of the same bench.
# Everything lands in gpurun_out/ (merged back by gpurun); summaries to keep are copied to profiles/ afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r1}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.txt 2>&1
echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu_$TAG.txt
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.stderr
echo "bench rc=$?"
cat gpurun_out/bench_$TAG.json; tail -20 gpurun_out/bench_$TAG.stderr
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/prof_$TAG.log 2>&1)
  echo "rocprof rc=$?"
  find /tmp/prof_$TAG -name "*stats*" | head
  for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$TAG.csv; done
  head -25 gpurun_out/rocprof_kernel_stats_$TAG.csv
  tail -3 /tmp/prof_$TAG.log
fi
