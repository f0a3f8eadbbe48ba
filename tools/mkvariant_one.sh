#!/bin/bash
# Build a variant of libfa_mi355.so that differs in ONE kernel file of csrc/ (the other objects are taken as built).
# usage: tools/mkvariant_one.sh <file stem, e.g. fa_decode_kernel> <name> "<extra flags>" -> tools/ab/lib_<name>.so
set -e
stem=$1; name=$2; shift 2
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/flash_attention_metal_amd/csrc
out=$root/tools/ab
mkdir -p $out/obj_$name
for f in $src/*.o; do b=$(basename $f .o); [ "$b" = "$stem" ] || cp $f $out/obj_$name/$b.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize $extra \
  -c $src/$stem.hip -o $out/obj_$name/$stem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/obj_$name/*.o
rm -rf $out/obj_$name
echo built $out/lib_$name.so
