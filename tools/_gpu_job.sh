#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 120 ./tools/probes/probe_launch_floor > gpurun_out/probe_launch_floor.log 2>&1
cat gpurun_out/probe_launch_floor.log
