#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "auto_routes or eight_wave or config3 or config2" > gpurun_out/routes_tests.log 2>&1 || { tail -40 gpurun_out/routes_tests.log; exit 1; }
tail -3 gpurun_out/routes_tests.log
timeout -k 10 600 python tools/auto_check.py > gpurun_out/auto_check.log 2>&1 || true
tail -30 gpurun_out/auto_check.log
