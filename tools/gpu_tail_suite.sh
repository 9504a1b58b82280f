#!/bin/bash
# the part of the GPU suite behind tests/test_gpu_parity.py's MAS tests (the full suite is ~19 min of box time: tools/gpu_suite.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { n=$1; shift; timeout $1 python -m pytest "${@:2}" -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/tail_$n.txt 2>&1; echo "$n rc=$? $(tail -1 gpurun_out/tail_$n.txt)"; grep -E "^FAILED|^ERROR" gpurun_out/tail_$n.txt | head -5; }
run mas 200 tests/test_gpu_parity.py tests/test_gpu_parity_full.py -k "mas or log_prior or compute_loss"
run training 300 tests/test_gpu_training.py
run parity 400 tests/test_gpu_parity.py -k "not mas"
run parity_full 400 tests/test_gpu_parity_full.py -k "not (mas or log_prior or compute_loss)"
