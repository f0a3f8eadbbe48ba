#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 1000 python tools/ab.py $L:10 tools/ab/lib_d128_k2.so:10 tools/ab/lib_d128_k4.so:10 tools/ab/lib_d128_v2.so:10 tools/ab/lib_d128_v4.so:10 --shapes c4,d128c4k --rounds 6 --iters 6 --warm-ms 600 > gpurun_out/ab_d128_knobs_warm.log 2>&1
cat gpurun_out/ab_d128_knobs_warm.log
