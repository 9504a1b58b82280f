#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE (cycles the GPU is busy) / kernel duration, old vs new conv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
for v in ws:libgradtts_gfx950.so old:libgtts_nows.so; do
  tag=${v%%:*}; lib=${v##*:}
  GTTS_STREAMS=0 GTTS_LIB=$ROOT/speech-backbones_amd/$lib timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d /tmp/clk_$tag -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline --no-roofline --no-extras --streams 0 > /tmp/clk_$tag.log 2>&1; echo "pass $tag rc=$?"
  ls /tmp/clk_$tag/*/ 2>/dev/null | head; 
  f=$(find /tmp/clk_$tag -name "*counter_collection.csv" | head -1); k=$(find /tmp/clk_$tag -name "*kernel_trace.csv" | head -1)
  python $ROOT/tools/clk_summarize.py "$f" "$k" > $ROOT/gpurun_out/clk_$tag.txt 2>&1; head -14 $ROOT/gpurun_out/clk_$tag.txt | cut -c1-200
done
