#!/bin/bash
# persistent convolution changes: parity subset (bit-exact batch invariance, taps, fenced build), then same-box A/B of variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_f16f8.py tests/test_gpu_fenced.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
OPS="^gtts::conv3x3_ws_kernel" bash tools/gpu_ab_ops.sh "$@"
