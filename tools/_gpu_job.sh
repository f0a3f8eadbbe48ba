#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 400 python tools/ab.py tools/ab/lib_fp8d128_v0.so:11 $L:11 --shapes c5d128,c5 --rounds 5 --iters 10 > gpurun_out/ab_fp8_d128b.log 2>&1
cat gpurun_out/ab_fp8_d128b.log
timeout -k 10 300 python3 tools/pmc.py gpurun_out/pmc_c5d128 2 16 8192 128 fp8 1 0 $L --script run_lib.py --iters 6 --sets sq1,sq2 > gpurun_out/pmc_c5d128.log 2>&1
python3 -c "
import json; r=json.load(open('gpurun_out/pmc_c5d128/pmc_summary.json')); print(r['derived'])"
