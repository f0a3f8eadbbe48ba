#!/bin/bash
# Evidence run for profiles/rNN (on the GPU box, via gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (counters only) for the dominant kernels, the C++ driver's full log. Output: gpurun_out/$1/
set -o pipefail
out=gpurun_out/${1:-prof}
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 200 --warmup 20 > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
echo "bench done"; tail -c 400 $out/bench_n1.json
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-sweep > $out/bench_under_rocprof.json 2> $out/rocprof_kt.err || exit 1
echo "kernel trace done"
python3 tools/pmc.py $out/pmc_c3 4 16 4096 64 bf16 1 auto --sets sq1,sq2,mem1,mem2 > $out/pmc_c3.log 2>&1 || exit 1
echo "pmc c3 done"
python3 tools/pmc.py $out/pmc_c4 1 32 16384 128 bf16 1 auto --iters 4 --sets sq1,sq2,mem1,mem2 > $out/pmc_c4.log 2>&1 || exit 1
echo "pmc c4 done"
python3 tools/pmc.py $out/pmc_c16k 1 64 16384 64 bf16 1 auto --iters 4 --sets sq1,sq2 > $out/pmc_c16k.log 2>&1 || exit 1
echo "pmc c16k done"
./driver/fa_driver --iters 10 > $out/driver_full.log 2>&1 || { echo "driver failed"; tail -5 $out/driver_full.log; exit 1; }
cp benchmark_results.csv benchmark_extended.csv $out/ 2>/dev/null
echo "driver done"; grep -E "PASSED|FAILED" $out/driver_full.log | head -12; grep -E "^c[2-5]" $out/driver_full.log
