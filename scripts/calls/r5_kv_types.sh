cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kv_types.py -m gpu -q 2>&1 | tail -3
for t in f16 q8_0 q4_0 q5_1 iq4_nl; do
  timeout 400 python bench.py --ctkv $t --steps 64 --no-cpu-baseline --pmc-traffic 0 2>/dev/null | tail -1 > gpurun_out/kv_$t.json
  python - "$t" <<'PY'
import json, sys
t = sys.argv[1]
j = json.loads(open(f"gpurun_out/kv_{t}.json").read())
k = j.get("kernel_classes_us", {})
print(t, j["value"], "tok/s", j["ms_per_step"], "ms/step prefill", j.get("prefill_tok_s"), {a: b for a, b in k.items() if "flash" in a or "kv_" in a or "set_rows" in a})
PY
done
