#!/bin/bash
# Build a variant of libfa_mi355.so with extra -D flags for the matrix-core kernels into tools/ab/lib_<name>.so
# (A/B material for tools/ab.py; tools/ab/ is git-ignored but travels to the GPU box).
# usage: tools/mkvariant.sh <name> "<extra flags>"
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flash_attention_metal_amd/csrc
out=$root/tools/ab
mkdir -p $out/obj_$name
common="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-division-by-zero"
for f in fa_api fa_scalar_kernels fa_bwd_kernels fa_fp8_kernel fa_decode_kernel; do
  [ -f $src/$f.o ] && cp $src/$f.o $out/obj_$name/$f.o
done
/opt/rocm/bin/hipcc $common -fno-honor-nans -fno-slp-vectorize $extra -c $src/fa_mfma_kernel.hip -o $out/obj_$name/fa_mfma_kernel.o &
/opt/rocm/bin/hipcc $common -fno-honor-nans -fno-slp-vectorize $extra -c $src/fa_mfma16_kernel.hip -o $out/obj_$name/fa_mfma16_kernel.o &
/opt/rocm/bin/hipcc $common -fno-honor-nans -fno-slp-vectorize $extra -c $src/fa_fwd_splitkv_kernel.hip -o $out/obj_$name/fa_fwd_splitkv_kernel.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/lib_$name.so
