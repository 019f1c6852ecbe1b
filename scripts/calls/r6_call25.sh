#!/bin/bash
# decode attention at 32 k context: split count x waves per workgroup (f16 cache), and the q8_0 cache
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { timeout 600 python bench.py "$@" --pmc-traffic 0 --no-cpu-baseline --timing-steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v for k, v in (d.get('kernel_classes_us') or {}).items() if 'attn' in k})"; }
for wv in 1 0; do for sp in 16 32 48 64; do
  echo "== f16 GGML_MI355X_FA_WV8=$wv GGML_MI355X_FA_SPLITS=$sp"; GGML_MI355X_FA_WV8=$wv GGML_MI355X_FA_SPLITS=$sp one --prefill 32000 --steps 48
done; done
for sp in 0 32 64; do echo "== q8_0 GGML_MI355X_FA_SPLITS=$sp"; GGML_MI355X_FA_SPLITS=$sp one --prefill 32000 --steps 48 --ctkv q8_0; done
