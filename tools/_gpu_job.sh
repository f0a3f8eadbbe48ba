#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 600 python tools/decode_ab.py $L tools/ab/lib_rs.so > gpurun_out/decode_ab_regstage.log 2>&1
cat gpurun_out/decode_ab_regstage.log
