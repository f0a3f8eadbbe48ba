#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -40 gpurun_out/gpu_tests_full.log; exit 1; }
tail -3 gpurun_out/gpu_tests_full.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_rw8.json 2> gpurun_out/bench_rw8.err
head -c 600 gpurun_out/bench_rw8.json; echo
