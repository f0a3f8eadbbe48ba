set -o pipefail
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -x -q -k "mfma16 or partial_last_key" > gpurun_out/r4g/tests_mfma16.log 2>&1; rc=$?; tail -5 gpurun_out/r4g/tests_mfma16.log; grep -n "Error\|assert " gpurun_out/r4g/tests_mfma16.log | head
L="tools/ab/lib_k_adds.so:4"; for n in v1 adds ones; do L="$L tools/ab/lib_k_$n.so:10"; done
python3 tools/ab.py $L --shapes c3,nc4k,nc8k,c16k,c8k,c2k,c1k --rounds 10 --iters 20 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > gpurun_out/r4g/ab_ones.log; cat gpurun_out/r4g/ab_ones.log
exit $rc
