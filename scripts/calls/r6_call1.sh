#!/bin/bash
# round 6, GPU call 1: the damped / peaked weight sets at full depth (north_star's bar as written), ggml-cpu's integer-dot arithmetic in the in-place
# quantised-KV decode kernels, the failed-exec-update path; then where a quantised cache stands at 2 k and 8 k context
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_log.txt
timeout 1500 python -m pytest tests/test_gpu_kv_types.py tests/test_gpu_model.py tests/test_gpu_full_depth.py -m gpu -x -q -p no:cacheprovider --durations=15 2>&1 | tail -32 | cut -c1-250
grep -h "parity-full-depth" gpurun_out/parity_log.txt | cut -c1-700
echo "== bench default"
timeout 600 python bench.py --steps 64 > gpurun_out/r6c1_bench.json 2> gpurun_out/r6c1_bench.err; tail -c 2500 gpurun_out/r6c1_bench.json; tail -3 gpurun_out/r6c1_bench.err
echo "== kv types x context"
for pre in 2048 7936; do for kv in f16 q8_0 q4_0 q5_1 iq4_nl; do
  timeout 300 python bench.py --steps 64 --warmup 8 --prefill $pre --ctkv $kv --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 > gpurun_out/r6c1_kv.json 2> gpurun_out/r6c1_kv.err
  python - "$pre" "$kv" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c1_kv.json').read().strip().splitlines()[-1])
    kc = d.get('kernel_classes_us', {})
    print(json.dumps({"prefill": int(sys.argv[1]), "ctkv": sys.argv[2], "tok_s": d['value'], "ms": d['ms_per_step'], "prefill_tok_s": d.get('prefill_tok_s'), "classes": {k: v for k, v in kc.items() if 'flash' in k or 'set_rows' in k or 'kv' in k or 'qkv' in k or 'rope' in k}}))
except Exception as e:
    print("ERR", sys.argv[1:], e, open('gpurun_out/r6c1_kv.err').read()[-400:])
PY
done; done | tee gpurun_out/r6c1_kv_types.jsonl
