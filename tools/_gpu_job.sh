#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 900 python tools/ab.py $L:10 tools/ab/lib_sumv.so:10 --shapes c3,c16k,nc4k,c4 --rounds 8 --iters 10 --warm-ms 600 > gpurun_out/ab_sumv.log 2>&1
cat gpurun_out/ab_sumv.log
