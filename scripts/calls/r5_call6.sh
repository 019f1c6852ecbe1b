#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "grown_cache or hipgraph" 2>&1 | tail -6 | cut -c1-300
grep "grown-cache" gpurun_out/parity_log.txt | tail -2
bash scripts/ab_env.sh GGML_MI355X_EXEC_UPDATE 0 1 -- --np 32 --prefill 128 2>&1 | cut -c1-110
timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 --np 32 --prefill 128 --steps 64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['graph_compute_host_us_per_step'])"
