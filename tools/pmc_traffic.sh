#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel-trace only), single stream.
# Calibration on this rocprofv3 / gfx950 (profiles/r01_pmc_fetch_v2.txt, known byte counts of prep_input, mul_mask,
# tail_identity, final_euler): bytes read = 2 x FETCH_SIZE KB, bytes written = WRITE_SIZE KB.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp GTTS_STREAMS=1
TAG=${1:-x}; ROOT=$PWD; cd /tmp
pass() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" > $ROOT/gpurun_out/pmc_${name}_$TAG.txt 2>&1; head -8 $ROOT/gpurun_out/pmc_${name}_$TAG.txt | cut -c1-120; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
