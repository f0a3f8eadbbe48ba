#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 300 python tools/ab.py $L:11 tools/ab/lib_hilo.so:11 --shapes c5,c5 --rounds 6 --iters 20 > gpurun_out/ab_fp8_hilo2.log 2>&1
cat gpurun_out/ab_fp8_hilo2.log
