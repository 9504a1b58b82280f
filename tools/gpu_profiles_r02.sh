#!/bin/bash
# Round-2 evidence on one GPU box visit (no parity tests here: see tools/gpu_final.sh): the bench lines of every workload,
# rocprofv3 kernel stats of the headline command (single stream), the HBM-traffic PMC passes and the SQ instruction-mix pass.
# Outputs in gpurun_out/ (copy what is to be kept into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}; ROOT=$PWD
timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.stderr; echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_$TAG.json
timeout 600 python bench.py --workload gradtts-multispk --precision bf16-store --timesteps 100 --steps 2 --no-cpu-baseline > gpurun_out/bench_multispk_bf16store_$TAG.json 2> gpurun_out/bench_multispk_bf16store_$TAG.stderr; echo "multispk rc=$?"
timeout 900 python bench.py --workload diffvc --vc-mode ml --timesteps 30 --steps 2 --no-cpu-baseline > gpurun_out/bench_diffvc_$TAG.json 2> gpurun_out/bench_diffvc_$TAG.stderr; echo "diffvc rc=$?"
timeout 600 python bench.py --workload hifigan --batch 16 --steps 2 > gpurun_out/bench_hifigan_$TAG.json 2> /dev/null; echo "hifigan rc=$?"
for g in "" "--graph"; do timeout 300 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras $g > gpurun_out/bench_B1${g}_$TAG.json 2> gpurun_out/bench_B1${g}_$TAG.stderr; done
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
