"""GPU parity of fa_fwd_decode (few query rows against a long key sequence; include/fa_mi355.h) against the fp64 oracle of the
generalised operator, and against fa_fwd_ex on the same inputs. Not in the reference (its operator is square and has one head
count, /root/reference/kernels.metal:606,619): unpinned by the reference like the rest of scope row f3."""
import numpy as np
import pytest

from util import LN2, effective_q, lse_tol, to_dev

pytestmark = pytest.mark.gpu

TOL_O = {"f16": 1.5e-3, "bf16": 6e-3}


@pytest.fixture(scope="module")
def fa():
    import torch

    import flash_attention_metal_amd as fa

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    fa.load_library()
    return fa


CASES = [  # B, Hq, Hkv, Nq, Nk, D, causal
    (1, 32, 8, 1, 1000, 64, True), (2, 32, 8, 1, 4096, 128, True), (1, 8, 8, 1, 65, 64, False), (4, 32, 8, 4, 700, 64, True),
    (1, 8, 1, 4, 513, 64, True), (1, 16, 2, 1, 2049, 128, False), (1, 4, 4, 16, 300, 64, True), (1, 8, 4, 16, 300, 128, True),
    (1, 2, 2, 32, 129, 64, True), (1, 64, 8, 1, 8192, 64, True), (1, 6, 3, 5, 77, 64, True), (3, 4, 2, 3, 1, 64, False),
    (1, 8, 2, 7, 7, 128, True), (1, 32, 4, 1, 16384, 128, True), (2, 16, 16, 2, 640, 64, False),
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_decode_vs_oracle_and_fwd_ex(fa, oracle_mod, dtype):
    import torch

    rng = np.random.default_rng(21)
    for (B, Hq, Hkv, Nq, Nk, D, causal) in CASES:
        q = oracle_mod.round_to(oracle_mod.init_random(B * Hq * Nq * D, int(rng.integers(1, 1 << 20))).reshape(B, Hq, Nq, D), dtype)
        k = oracle_mod.round_to(oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), dtype)
        v = oracle_mod.round_to(oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), dtype)
        qd, kd, vd = (to_dev(x, dtype) for x in (q, k, v))
        o, lse = fa.flash_attention_decode(qd, kd, vd, is_causal=causal)
        torch.cuda.synchronize()
        what = (B, Hq, Hkv, Nq, Nk, D, causal)
        on, ln = o.float().cpu().numpy(), lse.cpu().numpy()
        assert np.isfinite(on).all() and np.isfinite(ln).all(), what
        o64, l64 = oracle_mod.attn_fwd_ex_f64(q, k, v, causal)
        assert np.abs(on - o64).max() < TOL_O[dtype], (what, np.abs(on - o64).max())
        assert np.abs(ln - l64).max() < lse_tol(dtype, 1, q, k), what
        # strict: the exact operator on the pre-scaled operand the kernel multiplies
        o64q, l64q = oracle_mod.attn_fwd_ex_f64(effective_q(oracle_mod, q, dtype), k, v, causal, LN2)
        assert np.abs(on - o64q).max() < TOL_O[dtype] and np.abs(ln - l64q).max() < 1e-4, what
        # and the generalised forward on the same inputs (another kernel, another summation order: close, not equal)
        o2, l2 = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)
        assert (o.float() - o2.float()).abs().max().item() < 2 * TOL_O[dtype], what
        if causal and Nq > 1:  # bottom-right alignment: the first query of a head sees exactly Nk - Nq + 1 keys
            first = Nk - Nq + 1
            o1, _ = oracle_mod.attn_fwd_ex_f64(*(np.ascontiguousarray(x) for x in (q[:, :, :1], k[:, :, :first], v[:, :, :first])), False)
            assert np.abs(on[:, :, :1] - o1).max() < TOL_O[dtype], what


def test_decode_e4m3_inputs(fa, oracle_mod):
    """An e4m3 KV cache (and e4m3 queries: dtype fp8_e4m3, the forward's config-5 family; bf16 output): the tiles are widened exactly to
    bf16 on their way into LDS, so the arithmetic per tile is that of the bf16 decode on the widened tensors -- and the result holds the
    bf16 bar against the fp64 oracle on the e4m3 values."""
    import torch

    rng = np.random.default_rng(22)
    for (B, Hq, Hkv, Nq, Nk, D, causal) in CASES:
        if Nk * D % 16:  # e4m3 heads are 16-byte aligned
            continue
        q = oracle_mod.round_to(2.0 * oracle_mod.init_random(B * Hq * Nq * D, int(rng.integers(1, 1 << 20))).reshape(B, Hq, Nq, D), "fp8")
        k = oracle_mod.round_to(2.0 * oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), "fp8")
        v = oracle_mod.round_to(2.0 * oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), "fp8")
        what = (B, Hq, Hkv, Nq, Nk, D, causal)
        o8, l8 = fa.flash_attention_decode(to_dev(q, "fp8"), to_dev(k, "fp8"), to_dev(v, "fp8"), is_causal=causal)
        ob, lb = fa.flash_attention_decode(to_dev(q, "bf16"), to_dev(k, "bf16"), to_dev(v, "bf16"), is_causal=causal)
        torch.cuda.synchronize()
        # (the same arithmetic per tile as the bf16 path on the widened tensors; the e4m3 path splits the keys over twice the items,
        # so the partial results combine in another order: close, not equal)
        assert o8.dtype == torch.bfloat16 and (o8.float() - ob.float()).abs().max().item() < TOL_O["bf16"], what
        assert (l8 - lb).abs().max().item() < 2e-5, what
        o64, l64 = oracle_mod.attn_fwd_ex_f64(q, k, v, causal)
        assert np.abs(o8.float().cpu().numpy() - o64).max() < 2 * TOL_O["bf16"], what
        assert np.abs(l8.cpu().numpy() - l64).max() < lse_tol("bf16", 1, q, k), what
    # bf16 queries on an e4m3 KV cache (fa_fwd_decode_kv8): the serving layout; same arithmetic as the bf16 path on the widened cache
    for (B, Hq, Hkv, Nq, Nk, D, causal) in ((1, 32, 8, 1, 4096, 128, True), (2, 16, 4, 2, 1000, 64, True), (1, 8, 8, 4, 640, 64, False)):
        q = oracle_mod.round_to(oracle_mod.init_random(B * Hq * Nq * D, 5).reshape(B, Hq, Nq, D), "bf16")
        k = oracle_mod.round_to(2.0 * oracle_mod.init_random(B * Hkv * Nk * D, 6).reshape(B, Hkv, Nk, D), "fp8")
        v = oracle_mod.round_to(2.0 * oracle_mod.init_random(B * Hkv * Nk * D, 7).reshape(B, Hkv, Nk, D), "fp8")
        o8, l8 = fa.flash_attention_decode(to_dev(q, "bf16"), to_dev(k, "fp8"), to_dev(v, "fp8"), is_causal=causal)
        ob, lb = fa.flash_attention_decode(to_dev(q, "bf16"), to_dev(k, "bf16"), to_dev(v, "bf16"), is_causal=causal)
        torch.cuda.synchronize()
        assert o8.dtype == torch.bfloat16 and (o8.float() - ob.float()).abs().max().item() < TOL_O["bf16"] and (l8 - lb).abs().max().item() < 2e-5
        o64, l64 = oracle_mod.attn_fwd_ex_f64(q, k, v, causal)
        assert np.abs(o8.float().cpu().numpy() - o64).max() < 2 * TOL_O["bf16"]
        assert np.abs(l8.cpu().numpy() - l64).max() < lse_tol("bf16", 1, q, k)
    with pytest.raises(ValueError):  # f16 queries on an e4m3 cache: not offered (the arithmetic is bf16)
        fa.flash_attention_decode(to_dev(q, "f16"), to_dev(k, "fp8"), to_dev(v, "fp8"))
    # dtypes must agree; strides of an e4m3 tensor are multiples of 16
    x = torch.zeros(1, 8, 1, 64, dtype=torch.float8_e4m3fn, device="cuda")
    kv = torch.zeros(1, 8, 200, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        fa.flash_attention_decode(x, kv, kv)


def test_decode_known_answers_and_workspace(fa, oracle_mod):
    import torch

    # Q = 0 -> uniform softmax over the visible keys; V[j, 0] = delta(j, t): O[i, 0] = 1 / visible(i) if t is visible, EXACTLY 0 else.
    # t straddles tile (64) and split boundaries; Nq = 4 queries at the end of a 1000-key cache (bottom-right causal)
    B, Hq, Hkv, Nq, Nk, D = 1, 8, 2, 4, 1000, 64
    q = torch.zeros(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(B, Hkv, Nk, D, dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(fa.decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D), dtype=torch.uint8, device="cuda")
    for t in (0, 63, 64, 65, 511, 512, 995, 996, 997, 998, 999):
        v = torch.zeros(B, Hkv, Nk, D, dtype=torch.bfloat16, device="cuda")
        v[:, :, t, 0] = 1.0
        ws.fill_(0xFF)  # the workspace's previous contents are irrelevant (here: NaN patterns)
        o, lse = fa.flash_attention_decode(q, k, v, is_causal=True, workspace=ws)
        torch.cuda.synchronize()
        for iq in range(Nq):
            vis = Nk - Nq + iq + 1
            col = o[0, :, iq, 0].float().cpu().numpy()
            if t < vis:
                want = oracle_mod.round_to(np.full(Hq, 1.0 / vis, np.float32), "bf16")
                assert np.all(np.abs(col - want) <= 8e-3 * want), (t, iq)
            else:
                assert np.array_equal(col, np.zeros(Hq, np.float32)), (t, iq)
            assert np.abs(lse[0, :, iq].cpu().numpy() - np.log(vis)).max() < 1e-5
        assert np.count_nonzero(o[..., 1:].float().cpu().numpy()) == 0
    # deterministic bit for bit, and independent of what the workspace held
    v = torch.randn(B, Hkv, Nk, D, dtype=torch.bfloat16, device="cuda")
    qq = torch.randn(B, Hq, Nq, D, dtype=torch.bfloat16, device="cuda")
    o1, l1 = fa.flash_attention_decode(qq, k, v, is_causal=True, workspace=ws)
    ws.zero_()
    o2, l2 = fa.flash_attention_decode(qq, k, v, is_causal=True, workspace=ws)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    # errors are reported: too many packed rows, short workspace, causal with fewer keys than queries
    with pytest.raises(fa.FaError) as e:
        fa.flash_attention_decode(torch.zeros(1, 8, 16, 64, dtype=torch.bfloat16, device="cuda"), k[:, :1], k[:, :1])  # 8 * 16 rows
    assert e.value.status == -2
    with pytest.raises(fa.FaError) as e:
        fa.flash_attention_decode(qq, k, v, workspace=torch.empty(64, dtype=torch.uint8, device="cuda"))
    assert e.value.status == -1 and "workspace" in str(e.value)
    with pytest.raises(fa.FaError):
        fa.flash_attention_decode(torch.zeros(1, 2, 8, 64, dtype=torch.bfloat16, device="cuda"), k[:, :, :4], k[:, :, :4], is_causal=True)
