#!/bin/bash
# builds llama_box_amd/ab/wide_variants.so (and, with ASYNC=1, ab/wide_async.so): the product library with a PATCHED copy of csrc/mmq_skinny.hip —
# scripts/ubench/experiments/mmq_skinny_wide_variants.patch (round 4 prefill lab: GGML_MI355X_MMQ_WIDE_WG2 / _KH / _KO / GGML_MI355X_OCC_LOG) and
# mmq_skinny_wide_async.patch (the barrier-free step loop, GGML_MI355X_MMQ_WIDE_ASYNC=1).  The patches are against the product file of commit 41d53af;
# if it has moved on, `patch` says so.  A/B with scripts/ab_prefill.sh, or GGML_BACKEND_PATH=$PWD/llama_box_amd/ab/wide_variants.so bash scripts/wide_ko.sh
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}/llama_box_amd"
mkdir -p ab
F="--offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -I../include -Icsrc"
one() {  # one <patch name> <output .so>
  cp csrc/mmq_skinny.hip /tmp/$1.hip
  patch -s /tmp/$1.hip ../scripts/ubench/experiments/$1.patch
  /opt/rocm/bin/hipcc $F -c /tmp/$1.hip -o /tmp/$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$2 $(ls build/*.o | grep -v mmq_skinny.o) /tmp/$1.o -ldl -Wl,--no-undefined
}
one mmq_skinny_wide_variants wide_variants.so
[ -n "${ASYNC:-}" ] && one mmq_skinny_wide_async wide_async.so
cp libggml-mi355x.so ab/a_product.so
