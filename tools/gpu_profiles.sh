#!/bin/bash
# Per-round evidence on one GPU box visit (tools/gpu_profiles.sh r06): the driver's default bench line, rocprofv3 kernel statistics of the headline command (f16f8
# default, and the bf16x3 plan of rounds 1-4), of config 4 and of B = 1; HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# kernel-trace-only passes) of config 2 and config 4 in f16f8; the SQ instruction-mix pass; the per-op table.
# Outputs in gpurun_out/ (copy what is to be judged into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06}; ROOT=$PWD
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_tables_$TAG.txt; echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_$TAG.json
stats() { name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o prof -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras "$@" > /tmp/prof_$name.log 2>&1); echo "rocprof $name rc=$?"
  for f in $(find /tmp/prof_$name -name "*kernel_stats*.csv"); do cp $f gpurun_out/rocprof_kernel_stats_${name}_$TAG.csv; done
  head -4 gpurun_out/rocprof_kernel_stats_${name}_$TAG.csv | cut -c1-150; }
stats c2
GTTS_STREAMS=1 stats c2_bf16x3 --precision bf16x3
stats c4 --workload diffvc
stats b1 --batch 1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --per-op > /dev/null 2> gpurun_out/per_op_table_$TAG.txt
cd /tmp
pass() { name=$1; shift; extra="$1"; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras $extra > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/pmc_${name}_$TAG.txt 2>&1; head -3 $ROOT/gpurun_out/pmc_${name}_$TAG.txt | cut -c1-150; }
pass fetch "" FETCH_SIZE
pass write "" WRITE_SIZE
pass insts "" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass fetch_c4 "--workload diffvc" FETCH_SIZE
pass write_c4 "--workload diffvc" WRITE_SIZE
cd $ROOT
python tools/make_traffic_json.py gpurun_out/pmc_fetch_$TAG.txt gpurun_out/pmc_write_$TAG.txt gradtts 16 1024 gpurun_out/traffic_c2_$TAG.json
python tools/make_traffic_json.py gpurun_out/pmc_fetch_c4_$TAG.txt gpurun_out/pmc_write_c4_$TAG.txt diffvc 16 1024 gpurun_out/traffic_c4_$TAG.json
python - <<PY
import json
a = json.load(open("gpurun_out/traffic_c2_$TAG.json")); a["precision"] = "f16f8"; a["conv_ws"] = True
d = json.load(open("gpurun_out/traffic_c4_$TAG.json")); d["precision"] = "f16f8"; d["conv_ws"] = True
old = json.load(open("profiles/traffic.json"))
runs = [a, d] + [r for r in old.get("runs", []) if r.get("precision") != "f16f8"]
json.dump({"runs": runs}, open("gpurun_out/traffic_$TAG.json", "w"), indent=1)
print("traffic runs:", [(r.get("workload"), r.get("precision"), len(r["kernels"])) for r in runs])
PY
