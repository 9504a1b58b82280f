#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 200 python -c "
import importlib, json, torch
pkg = importlib.import_module('speech-backbones_amd')
c = pkg._lib.measured_ceilings(torch.device('cuda:0')); c.pop('hbm_detail')
print(json.dumps(c))" 2>&1 | grep -v amdgpu | tee gpurun_out/r04_ceilings3.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/hf_tr -o p -- python $ROOT/bench.py --workload hifigan --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/hf_tr.log 2>&1; echo "trace rc=$?"
f=$(find /tmp/hf_tr -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/hifigan_layers.py $f | tee $ROOT/gpurun_out/r04_hifigan_layers.txt
