#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 500 python tools/ab.py $L:10 tools/ab/lib_bn32o3.so:10 tools/ab/lib_bn32o2.so:10 --shapes c4,d128c8k,d128c4k,d128nc --rounds 5 --iters 10 > gpurun_out/ab_d128_bn32.log 2>&1
cat gpurun_out/ab_d128_bn32.log
