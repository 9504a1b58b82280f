#!/bin/bash
# Round-3 first GPU visit: parity subset on the wave-specialised conv + per-op A/B against the GTTS_WS=0 build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -q -x -p no:cacheprovider > gpurun_out/ws1_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/ws1_pytest.txt
for spec in "ws3 libgradtts_gfx950.so 3" "ws0 libgradtts_gfx950.so 0" "old3 libgtts_nows.so 3" "old0 libgtts_nows.so 0"; do
  set -- $spec
  GTTS_LIB=$PWD/speech-backbones_amd/$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --per-op --streams $3 > gpurun_out/ws1_$1.json 2> gpurun_out/ws1_$1.txt
  echo "== $1 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/ws1_$1.json'));print(d['value'], d['config'].get('ms_per_unet_call'), (d.get('roofline') or {}).get('avg_us'))")"
done
grep -E "conv3x3_ws|conv_mfma_kernel<0" gpurun_out/ws1_ws0.txt | head -40
