#!/bin/bash
# Builds a variant of libfsvit.so with one source recompiled under extra flags:
#   bash tools/build_variant.sh NAME file.hip "-DFOO=1 ..."   ->  tools/probes/variants/libfsvit_NAME.so
set -eu
name=$1; src=$2; flags=${3:-}
cd "$(dirname "$0")/../few-shot-vit_amd/csrc"
out=../../tools/probes/variants; mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $([ $src = mlp_rows.hip ] && echo -fno-slp-vectorize || true) $flags -c $src -o $out/${src%.hip}_$name.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfsvit_$name.so $objs $out/${src%.hip}_$name.o
echo "built $out/libfsvit_$name.so"
