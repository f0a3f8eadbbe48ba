set -o pipefail
mkdir -p gpurun_out/r4q
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4q/gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r4q/gpu_tests.log; grep -n "Error\|assert " gpurun_out/r4q/gpu_tests.log | head
exit $rc
