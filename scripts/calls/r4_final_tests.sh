#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; rm -f gpurun_out/parity_log.txt
timeout 1700 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45 > gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log
