#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; tail -n 12 gpurun_out/pytest_gpu_full.log
