set -o pipefail
mkdir -p gpurun_out/r4p
L="tools/ab/lib_d128.so:4 tools/ab/lib_d128.so:10"; for n in la3 la4 lak4 lav4; do L="$L tools/ab/lib_d_$n.so:10"; done
python3 tools/ab.py $L --shapes c4,d128nc --rounds 8 --iters 10 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > gpurun_out/r4p/ab_d128_knobs.log; cat gpurun_out/r4p/ab_d128_knobs.log
