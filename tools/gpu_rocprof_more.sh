#!/bin/bash
# rocprofv3 kernel-trace statistics of the configurations the headline profile does not cover: config 3 (bf16 storage), config 4
# (DiffVC) and B = 1 (single stream, one timed step each)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
run() { name=$1; shift
  (cd /tmp && GTTS_STREAMS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras "$@" > /tmp/prof_$name.log 2>&1); echo "rocprof $name rc=$?"
  for f in $(find /tmp/prof_$name -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$name.csv; done
  head -4 gpurun_out/rocprof_kernel_stats_$name.csv | cut -c1-150
}
run c3 --workload gradtts-multispk --precision bf16-store --timesteps 100
run c4 --workload diffvc --timesteps 6
run b1 --batch 1
