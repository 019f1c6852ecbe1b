#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for dec in 1 0; do
echo "== GGML_MI355X_FA_DEC64=$dec"
GGML_MI355X_FA_DEC64=$dec GGML_MI355X_FAKE_DEVICES=2 timeout 600 python tests/layer_split_worker.py 2>/dev/null | grep LAYER_SPLIT_JSON | python -c "
import sys, json
d = json.loads(sys.stdin.read()[len('LAYER_SPLIT_JSON '):])
for c in d['cases']:
    print(c['model'], 'fa', c['fa'], 'layers', c['n_layer'], 'prompt', c['n_prompt'], 'ub', c['n_ubatch'], '| one dev vs oracle %.2e' % c['nmse_one_device_vs_oracle'], '|', [(g, c['graphs%d' % g]['bit_equal_to_one_device'], '%.2e' % c['graphs%d' % g]['nmse_vs_one_device']) for g in (0, 1)])
"
done
