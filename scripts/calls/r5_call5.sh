#!/bin/bash
# round 5, GPU call 5: CPU leg placement A/B (interleaved + unpinned | node-local + pinned), then the secondary configurations with the new graph key
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== cpu: interleaved, unpinned"; LAYERS=8 THREADS=8,12,14,15,16 timeout 300 python scripts/cpu_scaling.py
echo "== cpu: node 0, pinned close"; GGML_LITE_NUMA_NODE=0 OMP_PROC_BIND=close OMP_PLACES=cores LAYERS=8 THREADS=8,12,14,15,16 timeout 300 python scripts/cpu_scaling.py
echo "== cpu: interleaved, pinned spread"; OMP_PROC_BIND=spread OMP_PLACES=cores LAYERS=8 THREADS=8,12,14,15,16 timeout 300 python scripts/cpu_scaling.py
echo "== cpu: node 0, pinned close, active wait"; GGML_LITE_NUMA_NODE=0 OMP_PROC_BIND=close OMP_PLACES=cores OMP_WAIT_POLICY=active LAYERS=8 THREADS=12,15,16 timeout 300 python scripts/cpu_scaling.py
echo "== configs"
rm -f gpurun_out/r5c5_configs.jsonl
for cfg in "--np 32 --prefill 128 --steps 64" "--np 8 --prefill 128 --steps 64" "--fa 0 --np 32 --prefill 128 --steps 64" "--fa 0"; do
  timeout 300 python bench.py --no-cpu-baseline --pmc-traffic 0 $cfg >> gpurun_out/r5c5_configs.jsonl 2>> gpurun_out/r5c5_configs.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r5c5_configs.jsonl'):
    d = json.loads(l)
    print(d['config']['workload'][:90], '|', d['value'], 'tok/s', d['ms_per_step'], 'ms | prefill', d.get('prefill_tok_s'), '| host', d.get('graph_compute_host_us_per_step'), d.get('hipGraphLaunch_host_us'))
    print('   ', {k: v for k, v in (d.get('kernel_classes_us') or {}).items()})
PY
