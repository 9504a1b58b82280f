#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_hifigan.py tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python bench.py --workload hifigan --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hifigan B16', d['ms_per_step'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/hf_tr -o p -- python $ROOT/bench.py --workload hifigan --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/hf_tr.log 2>&1; echo "trace rc=$?"
f=$(find /tmp/hf_tr -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/hifigan_layers.py $f > $ROOT/gpurun_out/r04_hifigan_layers_b.txt; tail -1 $ROOT/gpurun_out/r04_hifigan_layers_b.txt
