#!/bin/bash
# Round-3 evidence on one GPU box visit: rocprofv3 kernel stats of the headline command and of one training step, the
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate kernel-trace-only passes) of the headline AND of config 3
# (bf16 storage), the SQ instruction-mix pass, standalone bench lines of config 3 / B=1.  Outputs in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r03}; ROOT=$PWD
timeout 600 python bench.py --workload gradtts-multispk --precision bf16-store --timesteps 100 --steps 2 --no-cpu-baseline > gpurun_out/bench_multispk_bf16store_$TAG.json 2> gpurun_out/kernel_table_multispk_bf16store_$TAG.txt; echo "multispk rc=$?"
for w in 0 1; do timeout 300 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --conv-ws $w > gpurun_out/bench_B1_ws${w}_$TAG.json 2> gpurun_out/kernel_table_B1_ws${w}_$TAG.txt; echo "B1 ws$w rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_B1_ws${w}_$TAG.json'));print(d['value'], d['config']['ms_per_unet_call'])")"; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_nocpu_$TAG.json 2> gpurun_out/bench_nocpu_tables_$TAG.txt; echo "bench (no cpu baseline) rc=$? $(python -c "import json;d=json.load(open('gpurun_out/bench_nocpu_$TAG.json'));print(d['value'], d['extras']['config3_bf16_store']['ms_per_unet_call'], d['extras']['config2_conv_ws']['mel_frames_per_s'], d['extras']['config2_conv_ws']['roofline']['avg_us'])")"
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ws_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --conv-ws 1 > /tmp/prof_ws_$TAG.log 2>&1); echo "rocprof ws rc=$?"
for f in $(find /tmp/prof_ws_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_conv_ws_$TAG.csv; done
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_$TAG.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$TAG.csv; done
head -6 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-140
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr_$TAG -o prof -- python $ROOT/tools/train_prof.py > gpurun_out/train_steps_$TAG.txt 2>&1); echo "rocprof train rc=$?"
for f in $(find /tmp/prof_tr_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/train_step_rocprof_kernel_stats_$TAG.csv; done
python tools/stats_summary.py gpurun_out/train_step_rocprof_kernel_stats_$TAG.csv 3
export GTTS_STREAMS=1
cd /tmp
pass() { name=$1; shift; extra="$1"; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras $extra > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/pmc_${name}_$TAG.txt 2>&1; head -4 $ROOT/gpurun_out/pmc_${name}_$TAG.txt | cut -c1-150; }
pass fetch "" FETCH_SIZE
pass write "" WRITE_SIZE
pass insts "" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass fetch_ws "--conv-ws 1" FETCH_SIZE
pass write_ws "--conv-ws 1" WRITE_SIZE
pass insts_ws "--conv-ws 1" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass fetch_c3 "--workload gradtts-multispk --precision bf16-store" FETCH_SIZE
pass write_c3 "--workload gradtts-multispk --precision bf16-store" WRITE_SIZE
cd $ROOT
python tools/make_traffic_json.py gpurun_out/pmc_fetch_$TAG.txt gpurun_out/pmc_write_$TAG.txt gradtts 16 1024 gpurun_out/traffic_c2_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_ws_$TAG.txt gpurun_out/pmc_write_ws_$TAG.txt gradtts 16 1024 gpurun_out/traffic_c2ws_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_c3_$TAG.txt gpurun_out/pmc_write_c3_$TAG.txt gradtts-multispk 16 1024 gpurun_out/traffic_c3_$TAG.json
python - <<PY
import json
a = json.load(open("gpurun_out/traffic_c2_$TAG.json")); a["precision"] = "bf16x3"
b = json.load(open("gpurun_out/traffic_c3_$TAG.json")); b["precision"] = "bf16-store"
c = json.load(open("gpurun_out/traffic_c2ws_$TAG.json")); c["precision"] = "bf16x3"; c["conv_ws"] = True
json.dump({"runs": [a, c, b]}, open("gpurun_out/traffic_$TAG.json", "w"), indent=1)
print("traffic runs:", len(a["kernels"]), len(b["kernels"]))
PY
