#!/bin/bash
# round 5, visit L: the tests added / touched after the full-suite run of visit K
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_training.py tests/test_gpu_f16f8.py "tests/test_gpu_parity.py::test_large_ragged_batch_b32_t2048" tests/test_gpu_measure.py -m gpu -q -s -p no:cacheprovider -k "not n50_t1024" > gpurun_out/r05l_tests.txt 2>&1
tail -4 gpurun_out/r05l_tests.txt; grep -E "^FAILED|^ERROR|length_scale|have a different" gpurun_out/r05l_tests.txt | head -20
