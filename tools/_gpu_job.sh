#!/bin/bash
set -e
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err || { tail -20 gpurun_out/bench_c5.err; exit 1; }
python3 -c "
import json; r=json.loads(open('gpurun_out/bench_c5.json').read().strip().splitlines()[-1]); print(r['value'], r['roofline']['kernel'], r['c5_fp8'], r['c4_slice']['tflops_total'])"
