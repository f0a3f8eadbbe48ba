"""End to end on the GPU: the thin C++ host (driver/fa_driver, the main.mm replacement) through the C-ABI.
Checks what a user of the reference would look at: the verdict lines (main.mm:239-594), the CSV schema
(main.mm:604-605,873-876) as its plot_results.py consumes it, and the exit code (the reference returns 0
regardless, main.mm:1209; ours is non-zero on a failed check)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "driver", "fa_driver")
HEADER = "N,Naive(ms),Flash(ms),FlashV2(ms),FlashV3(ms),FlashV4(ms),SpeedupV1,SpeedupV2,SpeedupV3,SpeedupV4"


def run_driver(tmp_path, *args):
    assert os.path.exists(DRV), "build it with `make -C driver` (or __graft_entry__.build())"
    r = subprocess.run([DRV, *args], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    return r


@pytest.mark.parametrize("mode", [[], ["--same-qkv"]])  # independent Q,K,V (default) and the reference's Q == K == V
def test_driver_verdicts_csv_and_exit_code(tmp_path, mode):
    r = run_driver(tmp_path, "--sizes", "128,256,512", "--iters", "3", "--warmup", "1", "--no-cpu", "--no-configs", *mode)
    out = r.stdout
    assert r.returncode == 0, out[-2000:] + r.stderr[-2000:]
    for v in ("Naive Kernel PASSED", "V1 PASSED", "V2 PASSED", "V3 PASSED", "V4 PASSED", "CAUSAL PASSED",
              "V4 (bf16) PASSED", "CAUSAL (bf16) PASSED"):
        assert v in out, v
    assert "FAILED" not in out
    assert "--- Benchmarking ---" in out and "--- High Occupancy Benchmark (B=16, H=8) ---" in out
    lines = open(tmp_path / "benchmark_results.csv").read().splitlines()
    assert lines[0] == HEADER
    rows = [l.split(",") for l in lines[1:] if l]
    assert [int(x[0]) for x in rows] == [128, 256, 512] and all(len(x) == 10 for x in rows)
    assert all(float(x[j]) > 0 for x in rows for j in range(1, 6))
    # the high-occupancy table carries a measured backward time (main.mm:883 header)
    ho = out.split("SpeedupV4vsV2")[1].strip().splitlines()[:3]
    assert all(float(l.split(",")[3]) > 0 for l in ho)
    ext = open(tmp_path / "benchmark_extended.csv").read().splitlines()
    assert ext[0].startswith("N,kernel,dtype,causal,B,H,D,devices,median_ms") and len(ext) > 10


def test_driver_baseline_configs(tmp_path):
    r = run_driver(tmp_path, "--no-verify", "--no-sweep", "--no-high-occupancy", "--no-cpu", "--iters", "4")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    got = {l.split(",")[0]: l.split(",") for l in r.stdout.splitlines() if l[:2] in ("c2", "c3", "c4", "c5")}
    # config 2 also through the kernel BASELINE.json names for it ("V2-style tiled kernel") and through the 128-row
    # matrix-core kernel, next to "auto"
    assert set(got) == {"c2", "c2_v2", "c2_mfma", "c3", "c3_mfma32", "c4", "c5", "c5_bf16p", "c5_d128", "c5_d128_bf16p"}
    assert float(got["c2_v2"][8]) > float(got["c2"][8]) > 0  # median ms: the scalar V2 kernel is the slow one
    assert got["c3"][6] == "bf16" and got["c3"][7] == "1" and float(got["c3"][9]) > 100  # TFLOP/s
    assert got["c5"][6] == "fp8_e4m3"
