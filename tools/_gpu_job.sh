#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 900 python tools/ab.py $L:10 tools/ab/lib_ring3.so:10 --shapes c3,nc4k,nc2k,c3f16,c5k --rounds 10 --iters 12 --warm-ms 600 > gpurun_out/ab_ring3.log 2>&1
cat gpurun_out/ab_ring3.log
