#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "any_multiple or eight_wave or auto_routes" > gpurun_out/pad_tests.log 2>&1 || { tail -40 gpurun_out/pad_tests.log; exit 1; }
tail -3 gpurun_out/pad_tests.log
