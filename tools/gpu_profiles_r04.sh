#!/bin/bash
# Round-4 evidence on one GPU box visit: default bench (driver command), rocprofv3 kernel stats of the headline command (both
# convolution kernels), HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate kernel-trace-only passes) of config 2 (both
# kernels), config 3 (bf16 storage) AND config 4 (DiffVC), the SQ instruction-mix pass, one training step's kernel table,
# the un-traced stream timeline and the co-run probes.  Outputs in gpurun_out/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04}; ROOT=$PWD
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_tables_$TAG.txt; echo "bench rc=$?"; cut -c1-160 gpurun_out/bench_$TAG.json
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/prof_$TAG.log 2>&1); echo "rocprof rc=$?"
for f in $(find /tmp/prof_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_$TAG.csv; done
(cd /tmp && GTTS_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ws_$TAG -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --conv-ws 1 > /tmp/prof_ws_$TAG.log 2>&1); echo "rocprof ws rc=$?"
for f in $(find /tmp/prof_ws_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_conv_ws_$TAG.csv; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hf_$TAG -o prof -- python $ROOT/bench.py --workload hifigan --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/prof_hf_$TAG.log 2>&1); echo "rocprof hifigan rc=$?"
for f in $(find /tmp/prof_hf_$TAG -name "*kernel_stats*.csv"); do cp $f gpurun_out/hifigan_rocprof_kernel_stats_$TAG.csv; done
head -4 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-140
bash tools/gpu_train_table.sh
export GTTS_STREAMS=1
cd /tmp
pass() { name=$1; shift; extra="$1"; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras $extra > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/pmc_${name}_$TAG.txt 2>&1; head -3 $ROOT/gpurun_out/pmc_${name}_$TAG.txt | cut -c1-150; }
pass fetch "" FETCH_SIZE
pass write "" WRITE_SIZE
pass insts "" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass fetch_ws "--conv-ws 1" FETCH_SIZE
pass write_ws "--conv-ws 1" WRITE_SIZE
pass fetch_c3 "--workload gradtts-multispk --precision bf16-store" FETCH_SIZE
pass write_c3 "--workload gradtts-multispk --precision bf16-store" WRITE_SIZE
pass fetch_c4 "--workload diffvc" FETCH_SIZE
pass write_c4 "--workload diffvc" WRITE_SIZE
cd $ROOT
unset GTTS_STREAMS
python tools/make_traffic_json.py gpurun_out/pmc_fetch_$TAG.txt gpurun_out/pmc_write_$TAG.txt gradtts 16 1024 gpurun_out/traffic_c2_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_ws_$TAG.txt gpurun_out/pmc_write_ws_$TAG.txt gradtts 16 1024 gpurun_out/traffic_c2ws_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_c3_$TAG.txt gpurun_out/pmc_write_c3_$TAG.txt gradtts-multispk 16 1024 gpurun_out/traffic_c3_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_c4_$TAG.txt gpurun_out/pmc_write_c4_$TAG.txt diffvc 16 1024 gpurun_out/traffic_c4_$TAG.json
python - <<PY
import json
a = json.load(open("gpurun_out/traffic_c2_$TAG.json")); a["precision"] = "bf16x3"
b = json.load(open("gpurun_out/traffic_c3_$TAG.json")); b["precision"] = "bf16-store"
c = json.load(open("gpurun_out/traffic_c2ws_$TAG.json")); c["precision"] = "bf16x3"; c["conv_ws"] = True
d = json.load(open("gpurun_out/traffic_c4_$TAG.json")); d["precision"] = "bf16x3"
json.dump({"runs": [a, c, b, d]}, open("gpurun_out/traffic_$TAG.json", "w"), indent=1)
print("traffic runs:", len(a["kernels"]), len(b["kernels"]), len(d["kernels"]))
PY
for spec in "3 0" "2 1" "0 0"; do set -- $spec
timeout 300 python tools/timeline_untraced.py --streams $1 --conv-ws $2 2>&1 | grep -v amdgpu >> gpurun_out/timeline_untraced_$TAG.txt
done
timeout 300 python tools/corun_probe2.py 2>&1 | grep -v amdgpu > gpurun_out/corun_probe2_$TAG.txt
tail -3 gpurun_out/corun_probe2_$TAG.txt
