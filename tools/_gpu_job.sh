#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp8 or auto_routes" > gpurun_out/fp8_tests.log 2>&1 || { tail -40 gpurun_out/fp8_tests.log; exit 1; }
tail -3 gpurun_out/fp8_tests.log
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 400 python tools/ab.py $L:4 $L:11 --shapes c5d128,c5d128bf --rounds 5 --iters 10 > gpurun_out/ab_fp8_d128.log 2>&1 || true
cat gpurun_out/ab_fp8_d128.log
