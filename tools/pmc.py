#!/usr/bin/env python3
"""PMC counters of one shape of the operator, per launch (means over the profiled launches).
Runs `rocprofv3 --pmc ...` passes (counters only, never together with tracing) around tools/run_shape.py and
summarises the kernels whose name contains `--match` (default "fwd_").
usage: pmc.py OUTDIR B H N D dtype causal [variant] [--iters 6] [--match fwd_] [--sets all|sq|mem]
This process never touches the GPU itself; the profiled program is started directly behind `--`."""
import argparse, csv, glob, json, os, subprocess, sys

SETS = {
    "sq1": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"],
    "sq2": ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_COEXEC_CYCLES",
            "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"],
    "sq3": ["SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAVES",
            "SQ_INSTS_SMEM", "SQ_INST_LEVEL_LDS", "GRBM_GUI_ACTIVE"],
    "mem1": ["FETCH_SIZE", "GRBM_GUI_ACTIVE"],
    "mem2": ["WRITE_SIZE", "GRBM_GUI_ACTIVE"],
}
ap = argparse.ArgumentParser()
ap.add_argument("outdir"); ap.add_argument("shape", nargs="+")
ap.add_argument("--iters", type=int, default=6); ap.add_argument("--match", default="fwd_")
ap.add_argument("--sets", default="sq1,sq2")
ap.add_argument("--script", default="run_shape.py", help="program under tools/ to profile (run_bwd.py: shape = B H N dtype causal)")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(a.outdir, exist_ok=True)
nshape = 5 if a.script == "run_bwd.py" else 6
shape = a.shape[:nshape] + [str(a.iters)] + a.shape[nshape:]
res = {}
for name in a.sets.split(","):
    d = os.path.join(a.outdir, name)
    cmd = ["rocprofv3", "--pmc"] + SETS[name] + ["-d", d, "--output-format", "csv", "--", sys.executable,
                                                 os.path.join(root, "tools", a.script)] + shape
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    open(os.path.join(a.outdir, name + ".log"), "w").write(r.stdout)
    if r.returncode != 0:
        print(f"pass {name} failed rc={r.returncode}", file=sys.stderr)
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if a.match not in row["Kernel_Name"]:
                continue
            key = row["Counter_Name"]
            acc.setdefault(key, []).append(float(row["Counter_Value"]))
            res["kernel"] = row["Kernel_Name"][:80]
            res["vgpr"], res["agpr"] = row["VGPR_Count"], row["Accum_VGPR_Count"]
        for k, v in acc.items():
            v = v[3:] if len(v) > 4 else v  # drop the warm-up launches of run_shape.py
            res[k] = sum(v) / len(v)
if "GRBM_GUI_ACTIVE" in res:
    cyc = res["GRBM_GUI_ACTIVE"] / 8.0  # summed over 8 XCDs
    simd = 1024.0
    d = {"gpu_cycles_per_launch": cyc}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in res: d["mfma_busy_frac"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * simd)
    if "SQ_ACTIVE_INST_VALU" in res: d["valu_active_frac"] = 4 * res["SQ_ACTIVE_INST_VALU"] / (cyc * simd)
    if "SQ_INSTS_VALU" in res and "SQ_INSTS_MFMA" in res: d["valu_per_mfma"] = res["SQ_INSTS_VALU"] / res["SQ_INSTS_MFMA"]
    if "SQ_VALU_MFMA_COEXEC_CYCLES" in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res: d["coexec_over_mfma_busy"] = res["SQ_VALU_MFMA_COEXEC_CYCLES"] / res["SQ_VALU_MFMA_BUSY_CYCLES"]
    if "SQ_WAVE_CYCLES" in res:
        w = res["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in res: d[k.lower() + "_over_wave_cycles"] = res[k] / w
        d["wave_cycles_x4_over_simd_cycles"] = 4 * w / (cyc * simd)
    res["derived"] = d
json.dump(res, open(os.path.join(a.outdir, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
