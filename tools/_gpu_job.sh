#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 600 python tools/ab.py $L:11 tools/ab/lib_occ3.so:11 tools/ab/lib_hilo3.so:11 --shapes c5,c5 --rounds 8 --iters 10 --warm-ms 500 > gpurun_out/ab_fp8_hilo3.log 2>&1
cat gpurun_out/ab_fp8_hilo3.log
