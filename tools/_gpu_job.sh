#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 300 python tools/ab.py $L:10 tools/ab/lib_qhot.so:10 --shapes c3,c2k,c8k --rounds 6 --iters 20 > gpurun_out/ab_qhot.log 2>&1
cat gpurun_out/ab_qhot.log
