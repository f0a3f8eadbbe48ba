#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 400 python tools/ab.py tools/ab/lib_zz0.so:0 $L:0 tools/ab/lib_zz1.so:0 --shapes c1k,c2k,c3,c8k,c3x2,c3h,d128c2k,d128c4k,c5bf --rounds 6 --iters 20 > gpurun_out/ab_zz.log 2>&1
cat gpurun_out/ab_zz.log
