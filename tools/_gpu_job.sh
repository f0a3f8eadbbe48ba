set -o pipefail
mkdir -p gpurun_out/r4t
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/r4t/tests_decode.log 2>&1; rc=$?; tail -3 gpurun_out/r4t/tests_decode.log
timeout -k 10 300 python3 tools/decode_time.py 1 32 8 1 16384 64 1 32 32 1 16384 64 1 32 8 1 16384 128 1 32 32 1 16384 128 1 32 8 16 2048 128 4 32 8 1 8192 128 > gpurun_out/r4t/decode_time.log 2>&1; cat gpurun_out/r4t/decode_time.log
exit $rc
