#!/bin/bash
# LAB: builds variants of the backend library into llama_box_amd/ab/ for scripts/ab_decode.sh — only the translation units that include mmvq_types.h
# (mmvq.hip, qkv.hip) are rebuilt with the variant's flags; everything else is linked from llama_box_amd/build/.
#   usage: scripts/lab/build_ab.sh tag1 "-DFLAG..." [tag2 "-D..." ...]     (tag "base" with "" = the default build)
set -e
cd "$(dirname "$0")/../../llama_box_amd"
make -s -j4 libggml-mi355x.so
mkdir -p ab
HIPFLAGS="--offload-arch=gfx950 -fvisibility=hidden -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DGGML_MAX_NAME=128 -I../include"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_$tag
  for f in mmvq qkv; do /opt/rocm/bin/hipcc $HIPFLAGS $flags -c csrc/$f.hip -o build_$tag/$f.o & done; wait
  objs=""; for o in build/*.o; do b=$(basename $o); if [ -f build_$tag/$b ]; then objs="$objs build_$tag/$b"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$tag.so $objs -ldl -Wl,--no-undefined
  echo "built ab/$tag.so ($flags)"
done
ls -la ab/
