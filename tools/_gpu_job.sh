#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -40 gpurun_out/gpu_tests_full.log; exit 1; }
tail -2 gpurun_out/gpu_tests_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
