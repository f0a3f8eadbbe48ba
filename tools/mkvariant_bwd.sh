#!/bin/bash
# Build a variant of libfa_mi355.so whose BACKWARD kernels carry extra -D flags, into tools/ab/lib_<name>.so (for tools/ab_bwd.py).
# The other objects are the ones already built in csrc/ (run make there first).
# usage: tools/mkvariant_bwd.sh <name> "<extra flags>"
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flash_attention_metal_amd/csrc
out=$root/tools/ab
mkdir -p $out/obj_$name
for f in $src/*.o; do b=$(basename $f .o); [ "$b" = fa_bwd_kernels ] || cp $f $out/obj_$name/$b.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize $extra \
  -c $src/fa_bwd_kernels.hip -o $out/obj_$name/fa_bwd_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/lib_$name.so
