#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 1000 python tools/ab.py $L:10 tools/ab/lib_lak1.so:10 tools/ab/lib_lak3.so:10 tools/ab/lib_lav1.so:10 tools/ab/lib_lav3.so:10 tools/ab/lib_prio0.so:10 --shapes c3,c16k,nc4k --rounds 8 --iters 10 --warm-ms 600 > gpurun_out/ab_knobs_warm.log 2>&1
cat gpurun_out/ab_knobs_warm.log
