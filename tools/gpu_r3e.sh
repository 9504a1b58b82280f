#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_diffvc.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3e_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3e_tests.txt
timeout 300 python tools/ws_repro2.py 2>&1 | grep -v amdgpu.ids | grep -v "\[0, 0, 0\]"
for s in 2 3; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --streams $s > gpurun_out/r3e_h_s$s.json 2>/dev/null
  echo "headline streams $s: $(python -c "import json;d=json.load(open('gpurun_out/r3e_h_s$s.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
done
timeout 300 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --per-op > gpurun_out/r3e_b1.json 2> gpurun_out/r3e_b1.txt
echo "B=1: $(python -c "import json;d=json.load(open('gpurun_out/r3e_b1.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
timeout 300 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-roofline --graph > gpurun_out/r3e_b1g.json 2>/dev/null
echo "B=1 graph: $(python -c "import json;d=json.load(open('gpurun_out/r3e_b1g.json'));print(d['value'], d['config'].get('ms_per_unet_call'))")"
grep "\.conv  " gpurun_out/r3e_b1.txt | awk '{print $1, $2, $3, $4, $(NF-2)}'
