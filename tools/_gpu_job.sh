#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 ./tools/probes/probe_dq_atomic_floor > gpurun_out/probe_dq_atomic_floor.log 2>&1
cat gpurun_out/probe_dq_atomic_floor.log
