#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 300 python tools/ab.py tools/ab/lib_sums.so:11 $L:11 --shapes c5,c5 --rounds 6 --iters 20 > gpurun_out/ab_fp8_ones.log 2>&1
cat gpurun_out/ab_fp8_ones.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp8 or auto_routes" > gpurun_out/fp8_tests.log 2>&1 || { tail -40 gpurun_out/fp8_tests.log; exit 1; }
tail -3 gpurun_out/fp8_tests.log
