#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
DECODE_AB_FP8=1 timeout -k 10 600 python tools/decode_ab.py tools/ab/lib_kvd1.so $L > gpurun_out/decode_ab_kv8_depth.log 2>&1
cat gpurun_out/decode_ab_kv8_depth.log
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -x -q 2>&1 | tail -2
