#!/bin/bash
# cross-compiles the mmq probe binaries (one per MMQ_PROBE value) into scripts/ubench/build/ — they travel to the GPU box
cd "$(dirname "$0")"; mkdir -p build
for p in ${PROBES:-0 1 2 4 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGGML_MAX_NAME=128 -DMMQ_PROBE=$p ${EXTRA:-} -I../../include -I../../llama_box_amd/csrc mmq_probe.hip -o build/mmq_probe_${TAG:-}$p &
done
wait; ls -la build/
