#!/bin/bash
# shadow capture (a grown cache view's graph is captured behind the step's own eager launches): tests, -np 32 / -np 8 A/B on one box, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "grown_cache or failed_executable or replay or graph" 2>&1 | tail -3
one() { timeout 400 python bench.py "$@" --pmc-traffic 0 --no-cpu-baseline --timing-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('host'))"; }
for rep in 1 2; do for sh in 0 1; do
  echo "== GGML_MI355X_SHADOW_CAPTURE=$sh rep $rep: -np 32"; GGML_MI355X_SHADOW_CAPTURE=$sh one --np 32 --prefill 128 --steps 64
  echo "== GGML_MI355X_SHADOW_CAPTURE=$sh rep $rep: -np 8";  GGML_MI355X_SHADOW_CAPTURE=$sh one --np 8 --prefill 128 --steps 64
done; done
echo "== -np 32 --fa 0"; for sh in 0 1; do GGML_MI355X_SHADOW_CAPTURE=$sh one --np 32 --fa 0 --prefill 128 --steps 64; done
echo "== headline across a 256-cell boundary (prefill 2000, 128 steps)"; for sh in 0 1; do GGML_MI355X_SHADOW_CAPTURE=$sh one --prefill 2000 --steps 128; done
