#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -40 gpurun_out/gpu_tests_full.log; exit 1; }
tail -3 gpurun_out/gpu_tests_full.log
timeout -k 10 1500 bash tools/collect_profiles.sh r04f > gpurun_out/collect_r04f.log 2>&1 || { tail -30 gpurun_out/collect_r04f.log; exit 1; }
tail -30 gpurun_out/collect_r04f.log
