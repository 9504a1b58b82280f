#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -q -x -p no:cacheprovider -k "taps or golden or batch_size or config3 or odd" 2>&1 | tail -2
for rep in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/q_bench_$rep.json 2> gpurun_out/q_tables_$rep.txt
echo "bench: $(python -c "import json;d=json.load(open('gpurun_out/q_bench_$rep.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
grep -E "tail_identity" gpurun_out/q_tables_1.txt
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload gradtts-multispk --precision bf16-store --timesteps 100 > gpurun_out/q_bench_c3.json 2> gpurun_out/q_tables_c3.txt
echo "cfg3: $(python -c "import json;d=json.load(open('gpurun_out/q_bench_c3.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
grep -E "tail_identity|final_euler" gpurun_out/q_tables_c3.txt
