#!/bin/bash
# Memory-system PMC passes of the headline command (texture path / L1 / L2 counters per kernel; counters only with --kernel-trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TCC|TD|SQ|GRBM|SPI)_[A-Za-z0-9_]+" | sort -u > $ROOT/gpurun_out/pmc_avail.txt; wc -l $ROOT/gpurun_out/pmc_avail.txt
pass() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras > /tmp/pmc_$name.log 2>&1; echo "pass $name rc=$?"; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python $ROOT/tools/pmc_summarize.py "$f" 2>&1 | grep -E "^kernel|conv3x3_ws" > $ROOT/gpurun_out/pmc_mem_${name}.txt; cut -c1-260 $ROOT/gpurun_out/pmc_mem_${name}.txt; tail -3 /tmp/pmc_$name.log | cut -c1-200; }
pass a GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
pass b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pass c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass d TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TD_BUSY_sum
pass e SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
