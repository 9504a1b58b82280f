#!/bin/bash
# parity of both conv kernels after the plan option + the evidence passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python __graft_entry__.py --smoke > gpurun_out/r3g_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3g_smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_abi.py -m gpu -q -p no:cacheprovider > gpurun_out/r3g_pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3g_pytest.txt
bash tools/gpu_profiles_r03.sh r03
