cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_model.py -x -q -k "teacher_forced and not batch or failed_upload" 2>&1 | tail -3
for v in 0 1 0 1; do for cfg in "--preset llama3-70b-q4_k_m --prefill 128 --steps 24 --layers 16" "--preset llama3-70b-q4_k_m --prefill 128 --steps 24 --layers 16 --emulate-tp 8"; do
GGML_MI355X_QKV_PIPE=$v python bench.py $cfg --no-cpu-baseline --timing-steps 8 --pmc-traffic 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_classes_us']
print('pipe=$v %-40s %.4f ms/step | ' % ('$cfg'[-28:], d['ms_per_step']) + ' '.join('%s=%.2f' % (a.replace('qkv_fused_', 'qkv_'), b) for a, b in sorted(k.items()) if a.startswith('qkv')))"
done; done | tee gpurun_out/ab_qkv_pipe.txt
