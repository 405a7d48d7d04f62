#!/bin/bash
# Builds a variant of libfsvit.so with one source recompiled under extra flags:
#   bash tools/build_variant.sh NAME file.hip "-DFOO=1 ..."   ->  tools/probes/variants/libfsvit_NAME.so
# The timing-diagnostic switches (*_DIAG / *_CLK / *_NO_* / MR_PAD) are not in the shipped sources: when tools/probes/variants/<file>.diag.patch exists the
# variant is compiled from a patched COPY of the source (the tree is not modified).
set -eu
name=$1; src=$2; flags=${3:-}
cd "$(dirname "$0")/../few-shot-vit_amd/csrc"
out=../../tools/probes/variants; mkdir -p $out
in=$src
if [ -f $out/${src%.hip}.diag.patch ]; then
  in=${src%.hip}_diagsrc_$name.hip          # next to the original: the relative #includes keep working
  cp $src $in
  (sed "s|few-shot-vit_amd/csrc/$src|few-shot-vit_amd/csrc/$in|g" $out/${src%.hip}.diag.patch | (cd ../.. && git apply -)) || { rm -f $in; echo "the diag patch of $src no longer applies"; exit 1; }
  trap "rm -f $PWD/$in" EXIT
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $([ $src = mlp_rows.hip ] && echo -fno-slp-vectorize || true) $flags -c $in -o $out/${src%.hip}_$name.o
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfsvit_$name.so $objs $out/${src%.hip}_$name.o
echo "built $out/libfsvit_$name.so"
