#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -40 gpurun_out/gpu_tests_full.log; exit 1; }
tail -2 gpurun_out/gpu_tests_full.log
timeout -k 10 1500 bash tools/collect_profiles.sh r04g > gpurun_out/collect_r04g.log 2>&1 || { tail -30 gpurun_out/collect_r04g.log; exit 1; }
grep -E "^c[2-5]|bwd B4|PASSED|FAILED" gpurun_out/collect_r04g.log | tail -24
