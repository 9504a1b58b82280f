#!/bin/bash
# GPU tests of everything touched after the last full-suite run (vocoder launcher guard, MAS launcher, DiffVC modules, drop-in modules)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hifigan.py tests/test_gpu_diffvc.py tests/test_gpu_diffvc_training.py tests/test_gpu_fenced.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "not b32 and not full_size" > gpurun_out/subset_tests.txt 2>&1
tail -4 gpurun_out/subset_tests.txt; grep -E "^FAILED|^ERROR" gpurun_out/subset_tests.txt | head
