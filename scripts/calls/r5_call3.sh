#!/bin/bash
# round 5, GPU call 3: the rest of the suite behind the full-depth file (model / ops / e2e / split incl. -sm layer / tp incl. the failure drill), host facts, CPU thread scaling
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; rm -f gpurun_out/parity_log.txt
bash scripts/host_facts.sh 2>&1 | tee gpurun_out/r5_host_facts.txt
timeout 2400 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_parity_e2e.py tests/test_gpu_split.py tests/test_gpu_tp_p2p.py -m gpu -x -q --durations=12 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log | cut -c1-250
grep "parity-full-depth\|layer-split\|tp-p2p drill" gpurun_out/parity_log.txt | cut -c1-900
echo "== cpu scaling"
LAYERS=8 timeout 600 python scripts/cpu_scaling.py 2>&1 | tee gpurun_out/r5_cpu_scaling.txt
