#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "eight_wave" > gpurun_out/t8.log 2>&1 || { tail -40 gpurun_out/t8.log; exit 1; }
tail -3 gpurun_out/t8.log
