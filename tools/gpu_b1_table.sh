#!/bin/bash
# B = 1 (what Grad-TTS/inference.py runs): per-op HIP-event table, and eager vs hipGraph replay of the sampler call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-op > gpurun_out/b1.json 2> gpurun_out/b1_table.txt
echo "eager: $(python -c "import json;d=json.load(open('gpurun_out/b1.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --graph > gpurun_out/b1_graph.json 2>/dev/null
echo "graph: $(python -c "import json;d=json.load(open('gpurun_out/b1_graph.json'));print(d['value'], d['config']['ms_per_unet_call'])")"
cut -c1-150 gpurun_out/b1_table.txt | head -120
