"""(batch, head) sharding and the rank plumbing bench.py uses -- CPU only, incl. a world_size-2 gloo run."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_heads_is_a_partition():
    from flash_attention_metal_amd.shard import shard_heads

    for n in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_heads(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b and c <= d  # contiguous, disjoint, ordered
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1  # balanced
    # BASELINE config 4: 8*32 = 256 slices over 8 GPUs -> 32 each
    assert [shard_heads(256, 8, r) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        shard_heads(8, 2, 2)


WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["FA_ROOT"])
import torch
from flash_attention_metal_amd import ranks
info = ranks.init_ranks(use_gpu=False)           # gloo
lo, hi = ranks.my_slices(info, 64)               # bench.py: 64 (batch,head) slices per rank, weak scaling
ranks.barrier(info)
t0 = time.perf_counter()
elapsed = 0.010 * (info.rank + 1)                # rank r "takes" 10(r+1) ms
units = float(hi - lo) * 5                       # 5 steps over its slices
value, worst = ranks.aggregate_throughput(info, units, elapsed)
ranks.barrier(info)
out = {"rank": info.rank, "world": info.world, "backend": info.backend, "lo": lo, "hi": hi, "value": value, "worst": worst}
open(os.path.join(os.environ["FA_OUT"], f"rank{info.rank}.json"), "w").write(json.dumps(out))
ranks.finalize(info)
'''


def test_two_ranks_gloo_aggregate():
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, FA_ROOT=ROOT, FA_OUT=tmp, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=120) == 0
        res = [json.load(open(os.path.join(tmp, f"rank{r}.json"))) for r in range(2)]
    assert [r["backend"] for r in res] == ["gloo", "gloo"]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 64, 64, 128)  # disjoint, covering
    for r in res:  # every rank sees: total units of all ranks / the slowest rank's time
        assert r["worst"] == pytest.approx(0.020)
        assert r["value"] == pytest.approx((64 * 5 + 64 * 5) / 0.020)


def test_single_rank_needs_no_process_group():
    from flash_attention_metal_amd import ranks

    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        info = ranks.init_ranks(use_gpu=False)
        assert info.world == 1 and info.backend is None
        assert ranks.my_slices(info, 64) == (0, 64)
        assert ranks.aggregate_throughput(info, 10.0, 2.0) == (5.0, 2.0)
        ranks.barrier(info)
        ranks.finalize(info)
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v


def test_bench_entry_as_typed_spawns_ranks_without_a_launcher():
    # `python bench.py --gpus N` typed without torch.distributed.run must work. The parent touches
    # no GPU: it spawns N fresh children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set before they import torch), waits,
    # and relays rank 0's JSON line. Rehearsed here on CPU over gloo (the control plane is the same code).
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["MASTER_PORT"] = "29541"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--cpu-gloo-rehearsal"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(line) == 1  # ONE JSON line, from rank 0
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["backend"] == "gloo" and r["slices_rank0"] == [0, 64] and r["slices_total"] == 128
    assert r["units_per_s"] == pytest.approx((64 * 3 * 2) / 0.002)  # all ranks' units / the slowest rank's time
    # per-rank rates (the imbalance fields of the bench line): rank r takes r+1 ms for the same 64*3 units
    assert r["per_rank"]["units_per_s"] == pytest.approx([64 * 3 / 0.001, 64 * 3 / 0.002])
    assert r["per_rank"]["min"] == pytest.approx(64 * 3 / 0.002) and r["per_rank"]["max"] == pytest.approx(64 * 3 / 0.001)


def test_bench_parent_stops_the_other_ranks_when_one_dies():
    # a rank that dies before the first barrier must not leave the parent (and the surviving ranks) waiting for the
    # process-group timeout: FA_BENCH_FAIL_RANK makes rank 1 of the rehearsal exit at once
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_PORT="29547", FA_BENCH_FAIL_RANK="1")
    import time

    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--cpu-gloo-rehearsal"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and time.time() - t0 < 60


def test_bench_parent_process_never_imports_torch_or_the_library():
    # the spawn parent must not initialise a GPU: no module-level import of torch / the kernel library, and the
    # spawner itself neither touches torch nor execs anything
    src = open(os.path.join(ROOT, "bench.py")).read()
    top_imports = [l for l in src.splitlines() if l.startswith(("import ", "from "))]
    assert top_imports and not any("torch" in l or "flash_attention_metal_amd" in l for l in top_imports)
    body = src[src.index("def spawn_ranks"):src.index("def make_events")]
    code = "\n".join(l for l in body.splitlines() if not l.strip().startswith(("#", '"""')) and "imports neither" not in l)
    assert "import torch" not in code and "torch." not in code and "os.exec" not in code and "subprocess.Popen" in code
    main = src[src.index("def main()"):]
    assert main.index("return spawn_ranks(args)") < main.index("import torch")  # spawn decision comes first
