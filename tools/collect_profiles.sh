#!/bin/bash
# Evidence run for profiles/rNN (on the GPU box, via gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (counters only) for the dominant kernels, backward kernel stats, the C++ driver's full log, the GPU test log.
# Output: gpurun_out/$1/   (copy what is to be judged into profiles/rNN/)
set -o pipefail
out=gpurun_out/${1:-prof}
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_steps20_warmup5.json 2> $out/bench20.err || exit 1
echo "bench (driver flags) done"; head -c 300 $out/bench_n1_steps20_warmup5.json; echo
python3 bench.py --gpus 1 --steps 200 --warmup 20 > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
echo "bench (default) done"; head -c 300 $out/bench_n1.json; echo
rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-sweep > $out/bench_under_rocprof.json 2> $out/rocprof_kt.err || exit 1
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
echo "kernel trace done"
python3 tools/pmc.py $out/pmc_c3 4 16 4096 64 bf16 1 auto --sets sq1,sq2,mem1,mem2 > $out/pmc_c3.log 2>&1 || exit 1
cp $out/pmc_c3/pmc_summary.json $out/pmc_c3_summary.json
python3 - $out <<'PY'
import json, sys
out = sys.argv[1]
r = json.load(open(f"{out}/pmc_c3_summary.json"))
f, w = r["FETCH_SIZE"], r["WRITE_SIZE"]
json.dump({"bytes_per_launch": (2 * f + w) * 1024, "fetch_size_kib": f, "write_size_kib": w,
           "formula": "(FETCH_SIZE*2 + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide streaming reads, MI355X_MICROARCH.md HBM section); counts Infinity-Cache hits",
           "workload": "bench.py config 3 (B=4,H=16,N=4096,D=64,bf16,causal), kernel " + r.get("kernel", "?"),
           "source": "profiles/r04/pmc_c3_summary.json (separate rocprofv3 --pmc passes, tools/pmc.py, round 4 build; not measured in the bench run itself)",
           "algorithmic_bytes": 135266304}, open(f"{out}/hbm_traffic.json", "w"), indent=1)
PY
echo "pmc c3 done"
python3 tools/pmc.py $out/pmc_c4 1 32 16384 128 bf16 1 auto --iters 4 --sets sq1,sq2,mem1,mem2 > $out/pmc_c4.log 2>&1 || exit 1
cp $out/pmc_c4/pmc_summary.json $out/pmc_c4_summary.json
echo "pmc c4 done"
python3 tools/pmc.py $out/pmc_c16k 1 64 16384 64 bf16 1 auto --iters 4 --sets sq1,sq2 > $out/pmc_c16k.log 2>&1 || exit 1
cp $out/pmc_c16k/pmc_summary.json $out/pmc_c16k_summary.json
echo "pmc c16k done"
# the same heads at N = 8192 and the 32x32x16 kernel on config 3: per-tile / per-block costs come from the differences
python3 tools/pmc.py $out/pmc_c8k 1 64 8192 64 bf16 1 auto --iters 4 --sets sq1 > $out/pmc_c8k.log 2>&1 || exit 1
cp $out/pmc_c8k/pmc_summary.json $out/pmc_c8k_summary.json
python3 tools/pmc.py $out/pmc_c3_mfma32 4 16 4096 64 bf16 1 mfma --sets sq1,sq2 > $out/pmc_c3_mfma32.log 2>&1 || exit 1
cp $out/pmc_c3_mfma32/pmc_summary.json $out/pmc_c3_mfma32_summary.json
echo "pmc c8k / c3_mfma32 done"
# config 5 (fp8): the all-fp8 kernel
python3 tools/pmc.py $out/pmc_c5 4 16 8192 64 fp8 1 0 flash_attention_metal_amd/csrc/libfa_mi355.so --script run_lib.py --iters 6 --sets sq1,sq2 > $out/pmc_c5.log 2>&1 || exit 1
cp $out/pmc_c5/pmc_summary.json $out/pmc_c5_summary.json
echo "pmc c5 done"
python3 tools/decode_time.py > $out/decode_time.log 2>&1 || exit 1
echo "decode done"
# backward on the config-3 shape: kernel times and counters
python3 tools/run_bwd.py 4 16 4096 bf16 1 30 > $out/backward.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/kt_bwd --output-format csv -- python3 tools/run_bwd.py 4 16 4096 bf16 1 30 > /dev/null 2> $out/rocprof_bwd.err || exit 1
find $out/kt_bwd -name "*kernel_stats.csv" -exec cp {} $out/backward_kernel_stats.csv \;
python3 tools/pmc.py $out/pmc_bwd_dq 4 16 4096 bf16 1 --script run_bwd.py --match bwd_dq --sets sq1,sq2 > $out/pmc_bwd_dq.log 2>&1 || exit 1
cp $out/pmc_bwd_dq/pmc_summary.json $out/pmc_bwd_dq_summary.json
python3 tools/pmc.py $out/pmc_bwd_dkdv 4 16 4096 bf16 1 --script run_bwd.py --match bwd_dkdv --sets sq1,sq2 > $out/pmc_bwd_dkdv.log 2>&1 || exit 1
cp $out/pmc_bwd_dkdv/pmc_summary.json $out/pmc_bwd_dkdv_summary.json
echo "backward done"; cat $out/backward.log
./driver/fa_driver --iters 10 > $out/driver_full.log 2>&1 || { echo "driver failed"; tail -5 $out/driver_full.log; exit 1; }
cp benchmark_results.csv benchmark_extended.csv $out/ 2>/dev/null
echo "driver done"; grep -E "PASSED|FAILED" $out/driver_full.log | head -12; grep -E "^c[2-5]" $out/driver_full.log
rm -rf $out/kt $out/kt_bwd $out/pmc_*/sq* $out/pmc_*/mem*
