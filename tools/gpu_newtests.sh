#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_diffvc.py tests/test_distributed.py "tests/test_gpu_training.py::test_gradtts_compute_loss_multispeaker_gpu_vs_cpu" "tests/test_gpu_parity_full.py::test_config3_bf16_store_n100_free_running_mel_scale" -m gpu -q -s -p no:cacheprovider > gpurun_out/newtests.txt 2>&1; echo "pytest rc=$?"; grep -E "rel err|max\|err\||RCCL|passed|failed|Error|error" gpurun_out/newtests.txt | tail -30
