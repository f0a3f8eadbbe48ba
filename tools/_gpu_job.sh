#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/decode_tests.log 2>&1 || { tail -30 gpurun_out/decode_tests.log; exit 1; }
tail -2 gpurun_out/decode_tests.log
timeout -k 10 600 python tools/decode_time.py > gpurun_out/decode_time_final2.log 2>&1
tail -12 gpurun_out/decode_time_final2.log
