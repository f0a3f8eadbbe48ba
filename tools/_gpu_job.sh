#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/decode_tests.log 2>&1 || { tail -40 gpurun_out/decode_tests.log; exit 1; }
tail -3 gpurun_out/decode_tests.log
timeout -k 10 600 python tools/decode_time.py 1 32 8 1 4096 64  1 32 8 1 16384 64  1 32 8 1 16384 128  1 32 32 1 16384 64  1 32 32 1 16384 128  1 64 8 1 8192 128  4 32 8 1 8192 128  1 32 8 4 16384 128 > gpurun_out/decode_e4m3.log 2>&1
cat gpurun_out/decode_e4m3.log
