set -o pipefail
mkdir -p gpurun_out/r4n
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4n/gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r4n/gpu_tests.log; grep -n "Error\|assert " gpurun_out/r4n/gpu_tests.log | head
[ $rc -eq 0 ] || exit $rc
./driver/fa_driver --iters 10 > gpurun_out/r4n/driver_full.log 2>&1 || { echo "driver failed"; tail -5 gpurun_out/r4n/driver_full.log; exit 1; }
cp benchmark_results.csv benchmark_extended.csv gpurun_out/r4n/ 2>/dev/null
grep -E "PASSED|FAILED" gpurun_out/r4n/driver_full.log | head -12; grep -E "^c[2-5]" gpurun_out/r4n/driver_full.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4n/bench20.json 2> gpurun_out/r4n/bench20.err && head -c 700 gpurun_out/r4n/bench20.json && echo
