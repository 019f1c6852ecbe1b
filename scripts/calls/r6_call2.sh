#!/bin/bash
# round 6, GPU call 2: (1) what executed code size costs a short kernel (icache probe); (2) the block-format cache rows stored by the fused Q/K/V launch —
# byte-exactness tests, then tok/s by cache type x context again; (3) where the sporadic inner gaps of a replayed decode step sit
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== icache probe"; timeout 120 scripts/ubench/build/icache_probe 2>&1 | tee gpurun_out/r6c2_icache_probe.txt
echo "== kv tests"
timeout 900 python -m pytest tests/test_gpu_kv_types.py tests/test_gpu_ops.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | cut -c1-250
echo "== kv types x context (cache rows stored by the fused launch)"
for pre in 2048 7936; do for kv in f16 q8_0 q4_0 q5_1 iq4_nl bf16; do
  timeout 300 python bench.py --steps 64 --warmup 8 --prefill $pre --ctkv $kv --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 > gpurun_out/r6c2_kv.json 2> gpurun_out/r6c2_kv.err
  python - "$pre" "$kv" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c2_kv.json').read().strip().splitlines()[-1])
    kc = d.get('kernel_classes_us', {})
    print(json.dumps({"model": "llama3-8b-q4_k_m", "prefill": int(sys.argv[1]), "ctkv": sys.argv[2], "tok_s": d['value'], "ms": d['ms_per_step'], "prefill_tok_s": d.get('prefill_tok_s'), "classes": {k: v for k, v in kc.items() if 'flash' in k or 'set_rows' in k or 'kv' in k or 'qkv' in k or 'rope' in k}}))
except Exception as e:
    print("ERR", sys.argv[1:], e, open('gpurun_out/r6c2_kv.err').read()[-400:])
PY
done; done | tee gpurun_out/r6c2_kv_types.jsonl
for kv in f16 q8_0 q4_0; do
  timeout 300 python bench.py --preset qwen2-7b-q5_k_m --steps 64 --warmup 8 --prefill 7936 --ctkv $kv --no-cpu-baseline --pmc-traffic 0 --timing-steps 8 > gpurun_out/r6c2_kv.json 2> gpurun_out/r6c2_kv.err
  python - 7936 "$kv" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c2_kv.json').read().strip().splitlines()[-1])
    kc = d.get('kernel_classes_us', {})
    print(json.dumps({"model": "qwen2-7b-q5_k_m", "prefill": int(sys.argv[1]), "ctkv": sys.argv[2], "tok_s": d['value'], "ms": d['ms_per_step'], "prefill_tok_s": d.get('prefill_tok_s'), "classes": {k: v for k, v in kc.items() if 'flash' in k or 'set_rows' in k or 'kv' in k or 'qkv' in k or 'rope' in k}}))
except Exception as e:
    print("ERR", sys.argv[1:], e, open('gpurun_out/r6c2_kv.err').read()[-400:])
PY
done | tee -a gpurun_out/r6c2_kv_types.jsonl
echo "== timeline: where the inner gaps sit"
rm -rf gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o t --output-format csv -- python bench.py --steps 48 --warmup 4 --prefill 2048 --timing-steps 0 --no-cpu-baseline --pmc-traffic 0 > gpurun_out/r6c2_tl.json 2> gpurun_out/r6c2_tl.err
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/decode_gaps.py $f -w | tail -24 | cut -c1-600 | tee gpurun_out/r6c2_gaps.txt
[ -n "$f" ] && python scripts/decode_gaps.py $f -v | grep -E "^ +[0-9]" | head -60 | cut -c1-160 > gpurun_out/r6c2_step_timeline.txt
rm -rf gpurun_out/tl
