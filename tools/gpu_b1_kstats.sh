#!/bin/bash
# rocprofv3 kernel statistics of the B = 1 call for library variants: bash tools/gpu_b1_kstats.sh base small4 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
for v in "$@"; do
  if [ "$v" = base ]; then unset GTTS_LIB; else export GTTS_LIB=$ROOT/speech-backbones_amd/libgtts_$v.so; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --batch 1 > /tmp/prof_$v.log 2>&1); echo "rocprof $v rc=$?"
  for f in $(find /tmp/prof_$v -name "*kernel_stats*.csv"); do cp $f gpurun_out/b1_kstats_$v.csv; done
  grep "conv3x3_ws" gpurun_out/b1_kstats_$v.csv | cut -c1-140
done
