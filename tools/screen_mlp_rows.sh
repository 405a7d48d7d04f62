#!/bin/bash
# Code-placement screen of mlp_rows.hip: hand-placed wait states / counted waits that are one short show up as wrong tiles only
# at some instruction alignments.  Builds the kernel with 1..11 leading s_nops and runs the operator tests on each.
# The MR_PAD switch lives in tools/probes/variants/mlp_rows.diag.patch (applied to a copy of the source here).
# usage (on a GPU box, from the repo root): bash tools/screen_mlp_rows.sh
set -u
cd few-shot-vit_amd/csrc
cp build/mlp_rows.o /tmp/mlp_rows.o.keep
restore() {   # an interrupted run must not leave a padded kernel inside the product library
  cp /tmp/mlp_rows.o.keep build/mlp_rows.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfsvit.so build/*.o
}
cp mlp_rows.hip mlp_rows_pad.hip
sed 's|csrc/mlp_rows.hip|csrc/mlp_rows_pad.hip|g' ../../tools/probes/variants/mlp_rows.diag.patch | (cd ../.. && git apply -) || exit 1
trap 'restore; rm -f mlp_rows_pad.hip' EXIT INT TERM
for n in 1 2 3 5 7 11; do
  if [ -f ../../tools/probes/pad/mr_pad$n.o ]; then cp ../../tools/probes/pad/mr_pad$n.o build/mlp_rows.o
  else /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DMR_PAD=$n -c mlp_rows_pad.hip -o build/mlp_rows.o; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libfsvit.so build/*.o
  echo "MR_PAD=$n: $(cd ../..; python -m pytest tests/test_gpu_ops.py tests/test_gpu_soak.py -q -k "mlp_rows or vit_block_tail or linear_rows or qkv_attention_rows or patch_embed2x2" 2>&1 | tail -1)"
done
