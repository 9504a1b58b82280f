#!/bin/bash
# HiFi-GAN generator (B = 16 x 80 x 1024 mels): rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE / SQ passes (kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
CMD="python $ROOT/bench.py --workload hifigan --batch 16 --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hf_stats -o p -- $CMD > /tmp/hf0.log 2>&1; echo "stats rc=$?"
for f in $(find /tmp/hf_stats -name "*kernel_stats*.csv"); do cp $f $ROOT/gpurun_out/hifigan_rocprof_kernel_stats.csv; done
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/hf_$name -o p -- $CMD > /tmp/hf_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/hf_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/hifigan_pmc_$name.txt 2>&1; head -5 $ROOT/gpurun_out/hifigan_pmc_$name.txt | cut -c1-140; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
head -6 $ROOT/gpurun_out/hifigan_rocprof_kernel_stats.csv | cut -c1-150
