#!/bin/bash
# builds llama_box_amd/ab/wide_variants.so: the product library with scripts/ubench/experiments/mmq_skinny_wide_variants.hip in place of
# csrc/mmq_skinny.hip (round 4 prefill lab: GGML_MI355X_MMQ_WIDE_WG2 / _KH / _KO / GGML_MI355X_OCC_LOG).  A/B with scripts/ab_prefill.sh or
# GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/wide_variants.so bash scripts/wide_ko.sh
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}/llama_box_amd"
mkdir -p ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -I../include -Icsrc \
  -c ../scripts/ubench/experiments/mmq_skinny_wide_variants.hip -o /tmp/mmq_skinny_variants.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/wide_variants.so $(ls build/*.o | grep -v mmq_skinny.o) /tmp/mmq_skinny_variants.o -ldl -Wl,--no-undefined
# (experiments/mmq_skinny_wide_async.hip — the barrier-free step loop, GGML_MI355X_MMQ_WIDE_ASYNC=1 — builds the same way: SRC=... below)
if [ -n "${ASYNC:-}" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -I../include -Icsrc \
    -c ../scripts/ubench/experiments/mmq_skinny_wide_async.hip -o /tmp/mmq_skinny_async.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/wide_async.so $(ls build/*.o | grep -v mmq_skinny.o) /tmp/mmq_skinny_async.o -ldl -Wl,--no-undefined
fi
cp libggml-mi355x.so ab/a_product.so
