set -o pipefail
mkdir -p gpurun_out/r4l
L=""; for n in base occ5 occ3; do L="$L tools/ab/lib_f8_$n.so:11"; done
python3 tools/ab.py tools/ab/lib_f8_base.so:4 $L --shapes c5 --rounds 10 --iters 20 2>&1 | grep -v amdgpu.ids | tr '|' '\n' > gpurun_out/r4l/ab_fp8_occ.log; cat gpurun_out/r4l/ab_fp8_occ.log
