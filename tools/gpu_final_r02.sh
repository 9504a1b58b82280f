#!/bin/bash
# Round-2 final evidence on one GPU box visit: the full GPU test suite, the default bench line (with extras and CPU baseline),
# rocprofv3 kernel stats of the headline command (single stream), the HBM-traffic and instruction-mix PMC passes.
# Outputs in gpurun_out/ (copied into profiles/ afterwards).  SKIP_TESTS=1 skips the pytest part.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02f}; ROOT=$PWD
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.txt
fi
timeout 600 python bench.py --per-op > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.stderr; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_$TAG.json
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_$TAG.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$TAG.csv; done
head -6 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-140
export GTTS_STREAMS=1
cd /tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/pmc_${name}_$TAG.txt 2>&1; head -4 $ROOT/gpurun_out/pmc_${name}_$TAG.txt | cut -c1-150; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
cd $ROOT
python tools/make_traffic_json.py gpurun_out/pmc_fetch_$TAG.txt gpurun_out/pmc_write_$TAG.txt gradtts 16 1024 gpurun_out/traffic_$TAG.json
