set -o pipefail
mkdir -p gpurun_out/r4i
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4i/gpu_tests.log 2>&1; rc=$?; tail -6 gpurun_out/r4i/gpu_tests.log; grep -n "Error\|assert " gpurun_out/r4i/gpu_tests.log | head
[ $rc -eq 0 ] || exit $rc
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4i/bench20.json 2> gpurun_out/r4i/bench20.err && head -c 400 gpurun_out/r4i/bench20.json && echo
python3 tools/pmc.py gpurun_out/r4i/pmc_c3 4 16 4096 64 bf16 1 auto --sets sq1,sq2 > gpurun_out/r4i/pmc_c3.log 2>&1; grep -A14 '"derived"' gpurun_out/r4i/pmc_c3.log
