#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 1000 python tools/ab.py $L:10 tools/ab/lib_hg32.so:10 tools/ab/lib_hg16.so:10 --shapes c3,c8k,c3x2 --rounds 8 --iters 10 --warm-ms 600 > gpurun_out/ab_hg_warm.log 2>&1
cat gpurun_out/ab_hg_warm.log
