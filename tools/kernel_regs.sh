#!/bin/bash
# VGPR / AGPR / spill / LDS figures of every kernel in one object file of the build:  bash tools/kernel_regs.sh few-shot-vit_amd/csrc/build/X.o [name pattern]
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" /tmp/kr.fatbin || exit 1
$L/clang-offload-bundler --unbundle --type=o --input=/tmp/kr.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kr.co || exit 1
$L/llvm-readelf --notes /tmp/kr.co | grep -E "^\s+\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" | \
  awk '/\.name:/ {if (n) print n, v; n=$2; v=""} !/\.name:/ {v = v " " $1 $2} END {print n, v}' | c++filt | grep -E "${2:-.}"
