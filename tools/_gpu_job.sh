set -o pipefail
mkdir -p gpurun_out/r4u
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mfma16 or strongly_negative" > gpurun_out/r4u/tests.log 2>&1; rc=$?; tail -4 gpurun_out/r4u/tests.log; grep -n "Error\|assert " gpurun_out/r4u/tests.log | head
python3 tools/ab.py tools/ab/lib_prev.so:4 tools/ab/lib_prev.so:10 tools/ab/lib_ft.so:10 --shapes c3,c2k,c1k,c512,nc4k,c16k,c4 --rounds 10 --iters 20 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > gpurun_out/r4u/ab_first_tile.log; cat gpurun_out/r4u/ab_first_tile.log
exit $rc
