#!/bin/bash
# Verdict r3 item 4: L2 locality of the causal issue order on the SHIPPED (four-workgroup, clock-limited) 128-row kernel.
# Arms: head groups {all (0), 32, 16} compiled in (tools/mkvariant.sh hgN "-DFA_DEBUG_KNOBS -DFA_FORCE_HEAD_GROUP=N").
# (1) wall: interleaved in one process (tools/ab.py); (2) per arm, separate --pmc passes: FETCH_SIZE, cycles, effective clock.
set -o pipefail
out=gpurun_out/${1:-l2loc}
mkdir -p $out
export TMPDIR=/tmp
libs="tools/ab/lib_hg0.so:4 tools/ab/lib_hg32.so:4 tools/ab/lib_hg16.so:4"
python3 tools/ab.py $libs --shapes c3,c8k,c16k --rounds 10 --iters 20 > $out/ab_wall.log 2>&1 || { tail -5 $out/ab_wall.log; exit 1; }
cat $out/ab_wall.log
for hg in 0 32 16; do
  for shp in "4 16 4096 64" "4 16 8192 64"; do
    tag=hg${hg}_N$(echo $shp | cut -d' ' -f3)
    python3 tools/pmc.py $out/pmc_$tag $shp bf16 1 4 tools/ab/lib_hg$hg.so --script run_lib.py --iters 8 --sets sq1,mem1 > $out/pmc_$tag.log 2>&1 || { tail -5 $out/pmc_$tag.log; exit 1; }
    cp $out/pmc_$tag/pmc_summary.json $out/pmc_${tag}_summary.json
    rm -rf $out/pmc_$tag
    echo "pmc $tag done"
  done
done
python3 - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/pmc_hg*_summary.json")):
    r = json.load(open(f)); d = r.get("derived", {})
    print(os.path.basename(f), "FETCH_SIZE KiB", r.get("FETCH_SIZE"), "-> MB read (x2)", round(2 * r.get("FETCH_SIZE", 0) * 1024 / 1e6, 1),
          "cycles", round(d.get("gpu_cycles_per_launch", 0)), "mfma_busy", round(d.get("mfma_busy_frac", 0), 4))
PY
