#!/bin/bash
# quick sanity of the training path on one box visit: the training tests and four timed steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python tools/train_prof.py 2>&1 | grep step
timeout 300 python tools/train_prof.py 2>&1 | grep step
