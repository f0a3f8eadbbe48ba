#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/decode_tests.log 2>&1 || { tail -40 gpurun_out/decode_tests.log; exit 1; }
tail -2 gpurun_out/decode_tests.log
