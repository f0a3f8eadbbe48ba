#!/bin/bash
# Build a variant of libfa_mi355.so that differs only in csrc/fa_mfma16_kernel.hip (the file compiles in seconds):
# the other objects are taken from csrc/ as built. usage: tools/mkvariant16.sh <name> "<extra flags>" -> tools/ab/lib_<name>.so
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flash_attention_metal_amd/csrc
out=$root/tools/ab
mkdir -p $out/obj_$name
for f in fa_api fa_scalar_kernels fa_bwd_kernels fa_mfma_kernel fa_fp8_kernel fa_fwd_splitkv_kernel; do cp $src/$f.o $out/obj_$name/$f.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize $extra \
  -c $src/fa_mfma16_kernel.hip -o $out/obj_$name/fa_mfma16_kernel.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/lib_$name.so
