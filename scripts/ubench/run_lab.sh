#!/bin/bash
# builds and runs scripts/ubench/decode_lab.hip on the GPU box (the product objects travel with the snapshot in llama_box_amd/build/)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGGML_MAX_NAME=128 -Iinclude -c scripts/ubench/decode_lab.hip -o /tmp/decode_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -Iinclude -c scripts/ubench/experiments/mmvq_dma.hip -o /tmp/mmvq_dma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/decode_lab.o llama_box_amd/build/mmvq.o llama_box_amd/build/qkv.o llama_box_amd/build/fattn.o llama_box_amd/build/fattn_mma.o llama_box_amd/build/ops.o llama_box_amd/build/quantize.o /tmp/mmvq_dma.o -o /tmp/decode_lab
for d in ${DEPTHS:-4}; do echo "== GGML_MI355X_DMA_DEPTH=$d" | tee -a gpurun_out/decode_lab.txt; GGML_MI355X_DMA_DEPTH=$d timeout 300 /tmp/decode_lab 2>&1 | tee -a gpurun_out/decode_lab.txt; done
if [ "${STAMPS:-0}" = "1" ]; then
F="--offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DGGML_MAX_NAME=128 -Iinclude"
/opt/rocm/bin/hipcc $F -DMI_LAB_STAMPS -c -I llama_box_amd/csrc scripts/ubench/mmvq_stamped.hip -o /tmp/mmvq_st.o
/opt/rocm/bin/hipcc $F -DMI_LAB_STAMPS -c scripts/ubench/stamp_lab.hip -o /tmp/stamp_lab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/stamp_lab.o /tmp/mmvq_st.o -o /tmp/stamp_lab
timeout 300 /tmp/stamp_lab 2>&1 | tee gpurun_out/stamp_lab.txt
fi
if [ "${PROBE:-0}" = "1" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGGML_MAX_NAME=128 -Iinclude scripts/ubench/prologue_probe.hip -o /tmp/prologue_probe
timeout 120 /tmp/prologue_probe 2>&1 | tee gpurun_out/prologue_probe.txt
fi
if [ "${CHAIN:-0}" = "1" ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGGML_MAX_NAME=128 -Iinclude scripts/ubench/chain_probe.hip -o /tmp/chain_probe
timeout 60 /tmp/chain_probe 2>&1 | tee gpurun_out/chain_probe.txt
fi
