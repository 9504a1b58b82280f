#!/bin/bash
# Round-end evidence on one GPU box visit: parity tests, the three bench workloads, rocprofv3 kernel stats and the
# HBM-traffic PMC passes of the headline command.  Outputs in gpurun_out/ (copy what is to be kept into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-final}; ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.txt
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.stderr; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json | cut -c1-300
timeout 600 python bench.py --workload gradtts-multispk --precision bf16 --timesteps 100 --steps 2 > gpurun_out/bench_multispk_$TAG.json 2> gpurun_out/bench_multispk_$TAG.stderr; echo "multispk rc=$?"; cut -c1-200 gpurun_out/bench_multispk_$TAG.json
timeout 900 python bench.py --workload diffvc --vc-mode ml --timesteps 30 --steps 2 > gpurun_out/bench_diffvc_$TAG.json 2> gpurun_out/bench_diffvc_$TAG.stderr; echo "diffvc rc=$?"; cut -c1-200 gpurun_out/bench_diffvc_$TAG.json
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/prof_$TAG.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$TAG.csv; done
head -8 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-160
bash tools/pmc_traffic.sh $TAG
