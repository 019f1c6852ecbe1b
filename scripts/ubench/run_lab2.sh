#!/bin/bash
# builds and runs scripts/ubench/mmvq_lab2.hip on the GPU box (the product objects travel with the snapshot in llama_box_amd/build/)
#   VARIANTS="0 1"  -> experiments/mmvq_v2.hip with -DV2_VARIANT=n     KOS="0 1 2 3" -> experiments/mmvq_ko.hip with -DKO=n
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -Iinclude"
/opt/rocm/bin/hipcc $F -c scripts/ubench/mmvq_lab2.hip -o /tmp/mmvq_lab2.o
mkdir -p /tmp/labgen && python3 scripts/ubench/experiments/make_lab_copies.py /tmp/labgen  # mmvq_ko / _prio / _ring.hip from the CURRENT csrc/mmvq.hip
run() {  # $1 = tag, $2 = object
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/mmvq_lab2.o llama_box_amd/build/mmvq.o $2 -o /tmp/mmvq_lab2_bin
  echo "== $1" | tee -a gpurun_out/mmvq_lab2.txt
  timeout 300 /tmp/mmvq_lab2_bin ${LAB_OPS:-} 2>&1 | tee -a gpurun_out/mmvq_lab2.txt
}
for v in ${VARIANTS:-}; do
  /opt/rocm/bin/hipcc $F -fvisibility=hidden -DV2_VARIANT=$v -c scripts/ubench/experiments/mmvq_v2.hip -o /tmp/mmvq_v2_$v.o
  run "V2_VARIANT=$v" /tmp/mmvq_v2_$v.o
done
for d in ${RINGS:-}; do
  /opt/rocm/bin/hipcc $F -fvisibility=hidden -DRING_D=$d -I llama_box_amd/csrc -c /tmp/labgen/mmvq_ring.hip -o /tmp/mmvq_ring_$d.o
  run "RING_D=$d" /tmp/mmvq_ring_$d.o
done
for q in ${PRIOS:-}; do
  /opt/rocm/bin/hipcc $F -fvisibility=hidden -DLAB_PRIO=$q -I llama_box_amd/csrc -c /tmp/labgen/mmvq_prio.hip -o /tmp/mmvq_prio_$q.o
  run "LAB_PRIO=$q" /tmp/mmvq_prio_$q.o
done
for k in ${KOS:-}; do
  /opt/rocm/bin/hipcc $F -fvisibility=hidden -DKO=$k ${KO_DEFS:-} -I llama_box_amd/csrc -c /tmp/labgen/mmvq_ko.hip -o /tmp/mmvq_ko_$k.o
  run "KO=$k ${KO_DEFS:-}" /tmp/mmvq_ko_$k.o
done
