"""The C-ABI library: loads, exports every symbol include/fa_mi355.h declares,
answers the no-GPU queries, and rejects bad arguments before any launch. CPU only."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fa():
    import flash_attention_metal_amd as fa

    if not os.path.exists(fa.lib_path()):
        fa.build_library()
    return fa


def declared_functions():
    src = open(os.path.join(ROOT, "include", "fa_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(fa):
    names = declared_functions()
    assert "fa_fwd" in names and "fa_last_error" in names and len(names) >= 10
    lib = ctypes.CDLL(fa.lib_path())
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fa_mi355.h but not exported"
    from flash_attention_metal_amd._lib import SYMBOLS

    assert sorted(SYMBOLS) == names  # the Python binding covers the whole header


def test_version_and_names(fa):
    lib = fa.load_library()
    assert lib.fa_version() == 400
    assert [lib.fa_variant_name(i).decode() for i in range(12)] == ["auto", "naive", "tiled", "tiled_v2", "mfma", "mfma_pp", "mfma_splitkv", "mfma_split2", "mfma_exact", "mfma_h64s2", "mfma16", "mfma_fp8pv"]
    assert [lib.fa_dtype_name(i).decode() for i in range(4)] == ["f32", "f16", "bf16", "fp8_e4m3"]


def test_support_table(fa):
    for d in (64, 128):
        assert fa.supported("bf16", "mfma", d) and fa.supported("f16", "mfma", d)
        assert not fa.supported("bf16", "mfma_pp", d) and not fa.supported("fp8_e4m3", "mfma_pp", d)  # retired in version 400
        assert fa.supported("bf16", "mfma16", d) and fa.supported("f16", "mfma16", d) and not fa.supported("fp8_e4m3", "mfma16", d)
        assert not fa.supported("f32", "mfma", d) and not fa.supported("f32", "mfma16", d)
        assert fa.supported("fp8_e4m3", "mfma", d) and not fa.supported("fp8_e4m3", "tiled_v2", d)
        for v in ("naive", "tiled", "tiled_v2"):
            for t in ("f32", "f16", "bf16"):
                assert fa.supported(t, v, d)
    for d in (32, 96, 256):  # scope row f3: the other head dims run on the 128-row matrix-core kernel
        assert fa.supported("bf16", "mfma", d) and fa.supported("f16", "mfma", d) and fa.supported("bf16", "mfma16", d) == (d < 128)
    assert fa.supported("fp8_e4m3", "mfma", 256) and not fa.supported("fp8_e4m3", "mfma", 32) and not fa.supported("fp8_e4m3", "mfma", 96)
    assert not fa.supported("bf16", "mfma", 48) and not fa.supported("bf16", "mfma", 512)
    # the 16x16x32 kernel takes every multiple of 8 up to 128 (16-bit inputs): 64 / 128 natively, the others on zero-padded rows
    assert fa.supported("bf16", "mfma16", 48) and fa.supported("f16", "mfma16", 120) and fa.supported("bf16", "auto", 80)
    assert not fa.supported("bf16", "mfma16", 44) and not fa.supported("bf16", "mfma16", 136) and not fa.supported("fp8_e4m3", "mfma16", 48)
    assert fa.supported("fp8_e4m3", "mfma_fp8pv", 64) and fa.supported("fp8_e4m3", "mfma_fp8pv", 128) and not fa.supported("fp8_e4m3", "mfma_fp8pv", 256) and not fa.supported("bf16", "mfma_fp8pv", 64)
    assert fa.supported("bf16", "mfma_splitkv", 64) and not fa.supported("bf16", "mfma_splitkv", 128)  # head_dim 64 only since version 400
    lib = fa.load_library()
    assert lib.fa_resolve_variant(fa.DTYPES["bf16"], 64) == fa.VARIANTS["mfma"]
    assert lib.fa_resolve_variant(fa.DTYPES["bf16"], 96) == fa.VARIANTS["mfma"]
    assert lib.fa_resolve_variant(fa.DTYPES["f32"], 64) == fa.VARIANTS["tiled_v2"]
    assert lib.fa_resolve_variant(fa.DTYPES["f32"], 48) == -2


def test_algorithmic_work_matches_survey_8d(fa):
    # c3 = 1.3744e11 FLOP / 1.3527e8 B ; c2 = 2.1475e9 FLOP / 4.227e6 B
    assert fa.algorithmic_flops(4, 16, 4096, 64, True) == 2.0 * 64 * 4096 * 4096 * 64
    assert fa.algorithmic_bytes(4, 16, 4096, 64, "bf16") == 8 * 64 * 4096 * 64 + 4 * 64 * 4096
    assert fa.algorithmic_flops(1, 8, 1024, 64, False) == 4.0 * 8 * 1024 * 1024 * 64
    assert abs(fa.algorithmic_bytes(1, 8, 1024, 64, "f16") - 4.227e6) < 1e3
    assert fa.algorithmic_bytes(4, 16, 8192, 64, "fp8_e4m3") == (3 + 2) * 64 * 8192 * 64 + 4 * 64 * 8192


def test_bad_arguments_are_rejected_before_launch(fa):
    lib = fa.load_library()
    P = ctypes.c_void_p
    ok = P(0x1000)

    def call(q=ok, k=ok, v=ok, o=ok, B=1, H=1, N=128, D=64, scale=0.125, bs=None, hs=None, dtype=2, variant=0):
        hs = N * D if hs is None else hs
        bs = H * hs if bs is None else bs
        return lib.fa_fwd(q, k, v, o, None, B, H, N, D, scale, bs, hs, 0, dtype, variant, None)

    assert call(q=None) == -1 and b"null" in lib.fa_last_error()
    assert call(N=0) == -1
    assert call(scale=0.0) == -1 and b"scale" in lib.fa_last_error()
    assert call(dtype=9) == -1
    assert call(hs=100) == -1 and b"stride" in lib.fa_last_error()
    assert call(q=P(0x1004)) == -1 and b"aligned" in lib.fa_last_error()
    assert call(D=44) == -2 and b"no kernel" in lib.fa_last_error()  # (multiples of 8 up to 128 all have a kernel: padded rows)
    assert call(D=136) == -2 and call(D=48, dtype=3) == -2  # ... 16-bit inputs only, and nothing between 128 and 256
    assert call(dtype=0, variant=4) == -2 and b"mfma" in lib.fa_last_error()
    assert call(dtype=3, variant=3) == -2  # fp8 inputs exist for the matrix-core variant only
    assert call(dtype=3, hs=128 * 64 + 8) == -1  # fp8 heads must stay 16-byte aligned


def test_ex_and_bwd_reject_bad_strides_and_oversized_grids(fa):
    # fa_fwd_ex / fa_bwd carry the same overflow and stride guards as fa_fwd
    lib = fa.load_library()
    P = ctypes.c_void_p
    ok = P(0x1000)
    big = 1 << 30
    # B*Hq*ceil(Nq/128) must fit an int
    assert lib.fa_fwd_ex(ok, ok, ok, ok, None, 65536, 65536, 65536, 128, 128, 64, 0.125, 65536 * 128 * 64, 128 * 64,
                         65536 * 128 * 64, 128 * 64, 0, 2, None) == -1 and b"grid" in lib.fa_last_error()
    assert lib.fa_fwd_ex(ok, ok, ok, ok, None, 2, 2, 2, 128, 128, 64, 0.125, -16384, 8192, 16384, 8192, 0, 2, None) == -1
    # one head of 4 GiB minus one tile: refused with the same +128-row margin as fa_fwd (the staging loops address past the end)
    n_big = (1 << 32) // (64 * 2) - 64
    assert lib.fa_fwd_ex(ok, ok, ok, ok, None, 1, 1, 1, 128, n_big, 64, 0.125, 128 * 64, 128 * 64, n_big * 64, n_big * 64, 0, 2, None) == -1 \
        and b"4 GiB" in lib.fa_last_error()
    # fa_fwd_exv: kernels that do not take the generalised problem are refused by name (no silent substitute)
    assert lib.fa_fwd_exv(ok, ok, ok, ok, None, 1, 4, 2, 128, 128, 64, 0.125, 4 * 8192, 8192, 2 * 8192, 8192, 0, 2, 5, None) == -2 \
        and b"mfma_pp" in lib.fa_last_error()
    f = P(0x2000)
    assert lib.fa_bwd(ok, ok, ok, ok, ok, f, f, f, f, f, 65536, 65536, 128, 64, 0.125, 65536 * 8192, 8192, 0, 2, None) == -1 \
        and b"grid" in lib.fa_last_error()
    assert lib.fa_bwd(ok, ok, ok, ok, ok, f, f, f, f, f, 2, 2, 128, 64, 0.125, -16384, 8192, 0, 2, None) == -1
    # grouped-query backward: Hkv must divide Hq; the key/value strides are checked like the query's
    assert lib.fa_bwd_ex(ok, ok, ok, ok, ok, f, f, f, f, f, 2, 6, 4, 128, 128, 64, 0.125, 6 * 8192, 8192, 4 * 8192, 8192, 0, 2, None) == -1 \
        and b"Hkv" in lib.fa_last_error()
    assert lib.fa_bwd_ex(ok, ok, ok, ok, ok, f, f, f, f, f, 2, 4, 2, 128, 128, 64, 0.125, 4 * 8192, 8192, 2 * 8192, 100, 0, 2, None) == -1 \
        and b"key/value strides" in lib.fa_last_error()
    # causal with fewer keys than queries would leave empty rows (as in fa_fwd_ex)
    assert lib.fa_bwd_ex(ok, ok, ok, ok, ok, f, f, f, f, f, 1, 2, 2, 256, 128, 64, 0.125, 2 * 16384, 16384, 2 * 8192, 8192, 1, 2, None) == -2
    assert big > 0


def test_operator_refuses_cpu_tensors(fa):
    import torch

    x = torch.zeros(1, 1, 128, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        fa.flash_attention_forward(x, x, x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        fa.ForwardPlan(x, x, x)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing in the package or the driver may reference it
    bad = []
    for d in ("flash_attention_metal_amd", "driver", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, d)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".cpp", ".c", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"import oracle|from oracle|oracle/|libfa_oracle|attn_oracle", txt):
                        # docstrings may MENTION that the oracle lives in oracle/: allow only that phrase
                        for line in txt.splitlines():
                            if re.search(r"import oracle|from oracle|libfa_oracle|attn_oracle\.h", line):
                                bad.append((fn, line.strip()))
    assert not bad, bad
