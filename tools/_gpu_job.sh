#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_full.log 2>&1 || { tail -40 gpurun_out/gpu_tests_full.log; exit 1; }
tail -2 gpurun_out/gpu_tests_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_last.json 2> gpurun_out/bench_last.err
python3 -c "
import json; r=json.loads(open('gpurun_out/bench_last.json').read().strip().splitlines()[-1]); print(r['value'], r['roofline']['achieved'], r['roofline']['kernel'], r['c5_fp8']['tflops'], r['c4_slice']['tflops_total'])"
