set -o pipefail
mkdir -p gpurun_out/r4b
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma16" > gpurun_out/r4b/tests_mfma16.log 2>&1; rc=$?; tail -5 gpurun_out/r4b/tests_mfma16.log; [ $rc -eq 0 ] || exit $rc
python3 tools/ab.py tools/ab/lib_r4b.so:4 tools/ab/lib_r4b.so:10 --shapes c3,nc4k,c8k,nc8k,c16k,c2k,c1k --rounds 10 --iters 20 > gpurun_out/r4b/ab_mfma16.log 2>&1; cat gpurun_out/r4b/ab_mfma16.log
