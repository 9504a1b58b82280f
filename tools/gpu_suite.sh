#!/bin/bash
# the round-end GPU checks the driver runs: full `pytest -m gpu`, then __graft_entry__.smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
TAG=${1:-r04}
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$TAG.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee -a gpurun_out/pytest_gpu_$TAG.txt
