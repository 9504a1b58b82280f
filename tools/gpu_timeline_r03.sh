#!/bin/bash
# kernels-in-flight statistics (tools/timeline.py) of the default sampler call with either convolution kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
for v in 0 1; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --conv-ws $v > /tmp/tl_$v.log 2>&1); echo "trace ws=$v rc=$?"
  f=$(find /tmp/tl_$v -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $f > gpurun_out/timeline_ws$v.txt 2>&1; head -7 gpurun_out/timeline_ws$v.txt
done
