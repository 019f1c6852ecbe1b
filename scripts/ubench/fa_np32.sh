# attention of a -np 32 decode step under the split / list knobs (kernel classes from the eager timing pass)
run() { echo "== $*"; env "$@" timeout 300 python bench.py --preset llama3-8b-q4_k_m --np 32 --prefill 128 --steps 32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['kernel_classes_us'].items() if 'attn' in k or 'rope' in k})"; }
run A=1
run GGML_MI355X_FA_SPLITS=1
run GGML_MI355X_FA_SPLITS=2
run GGML_MI355X_FA_SPLITS=6
run GGML_MI355X_FA_LIST=0
run GGML_MI355X_FA_MMA_MIN_Q=32
