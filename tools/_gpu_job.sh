#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -x -q > gpurun_out/bwd_tests.log 2>&1 || { tail -40 gpurun_out/bwd_tests.log; exit 1; }
tail -5 gpurun_out/bwd_tests.log
timeout -k 10 200 python tools/run_bwd.py 4 16 4096 bf16 1 30 > gpurun_out/bwd_time.log 2>&1
tail -12 gpurun_out/bwd_time.log
