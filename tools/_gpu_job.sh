#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 600 python tools/ab.py tools/ab/lib_fp8_mov.so:11 $L:11 --shapes c5,c5d128,c5 --rounds 8 --iters 10 --warm-ms 500 > gpurun_out/ab_fp8_nomov.log 2>&1
cat gpurun_out/ab_fp8_nomov.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fp8" 2>&1 | tail -2
