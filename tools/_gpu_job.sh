#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q > gpurun_out/bwd_tests.log 2>&1 || { tail -40 gpurun_out/bwd_tests.log; exit 1; }
tail -3 gpurun_out/bwd_tests.log
