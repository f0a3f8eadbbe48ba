#!/bin/bash
set -e
mkdir -p gpurun_out
L=flash_attention_metal_amd/csrc/libfa_mi355.so
timeout -k 10 900 python tools/ab_bwd.py $L tools/ab/lib_bwd_la2.so tools/ab/lib_bwd_la4.so tools/ab/lib_bwd_lb1.so tools/ab/lib_bwd_lb3.so --shapes c3,nc4k --rounds 8 --iters 4 > gpurun_out/ab_bwd_knobs_warm.log 2>&1
cat gpurun_out/ab_bwd_knobs_warm.log
