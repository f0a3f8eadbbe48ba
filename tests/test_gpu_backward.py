"""GPU parity of the backward pass (scope row f1), through the C-ABI, against the fp64 CPU oracle.

Reference: /root/reference/kernels.metal:905-1265. The reference's own CPU check of it is broken
(main.mm:1100-1101 value-casts bit patterns; dK/dV are never compared), so backward parity is
UNPINNED by the reference: the anchor is oracle_attn_bwd_f64 (cross-checked against torch autograd
in fp64 by tests/test_oracle_backward.py). Tolerance: max|g - g_ref| / max|g_ref| per gradient tensor,
  f16: 4e-3, bf16: 2e-2  (P and dS enter the second products rounded to 16 bits; the reference's bar is 1e-1, main.mm:1191).
"""
import numpy as np
import pytest

from util import make_qkv, to_dev

pytestmark = pytest.mark.gpu
TOL = {"f16": 4e-3, "bf16": 2e-2}


@pytest.fixture(scope="module")
def fa():
    import torch

    import flash_attention_metal_amd as fa

    assert torch.cuda.is_available()
    fa.load_library()
    return fa


def grads(fa, q, k, v, do, dtype, causal):
    import torch

    qd, kd, vd, dod = (to_dev(x, dtype) for x in (q, k, v, do))
    o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)
    dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o, dod, lse, is_causal=causal)
    torch.cuda.synchronize()
    return dq.cpu().numpy(), dk.cpu().numpy(), dv.cpu().numpy()


def rel(a, ref):
    # relative to the largest reference gradient; a gradient that is exactly zero in exact arithmetic (a single
    # key: P = 1, dS = P (dP - delta) = 0) carries the fp32 rounding of dP (matrix core) against delta (FMA chain in
    # the dQ kernel's prologue): a few 1e-8 on values of order 1, hence the absolute floor
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-4)


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_vs_oracle(fa, oracle_mod, dtype, causal, D):
    for (B, H, N) in ((1, 1, 128), (2, 3, 200), (1, 2, 65), (1, 1, 1), (1, 1, 63), (1, 2, 129), (2, 2, 520)):
        q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
        do = oracle_mod.round_to(oracle_mod.init_random(B * H * N * D, 45).reshape(B, H, N, D), dtype)
        dq, dk, dv = grads(fa, q, k, v, do, dtype, causal)
        rq, rk, rv = oracle_mod.attn_bwd_f64(q, k, v, do, causal)
        for name, g, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
            assert np.isfinite(g).all(), (name, B, H, N)
            assert rel(g, ref) < TOL[dtype], (name, dtype, causal, B, H, N, rel(g, ref))


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_grouped_query_heads(fa, oracle_mod, causal, D):
    """fa_bwd_ex (scope rows f1 + f3): query head h reads key/value head h // G; dK / dV of a key head are the sums over its
    G query heads. Oracle: the fp64 backward on K / V repeated to Hq heads, its dK / dV summed per group."""
    import torch

    dtype = "bf16"
    for (B, Hq, Hkv, N) in ((1, 4, 2, 200), (2, 6, 1, 129), (1, 8, 4, 64), (1, 3, 3, 130)):
        G = Hq // Hkv
        q, _, _ = make_qkv(oracle_mod, B, Hq, N, D, dtype)
        _, k, v = make_qkv(oracle_mod, B, Hkv, N, D, dtype)
        do = oracle_mod.round_to(oracle_mod.init_random(B * Hq * N * D, 45).reshape(B, Hq, N, D), dtype)
        qd, kd, vd, dod = (to_dev(x, dtype) for x in (q, k, v, do))
        o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)
        dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o, dod, lse, is_causal=causal)
        torch.cuda.synchronize()
        assert dq.shape == qd.shape and dk.shape == kd.shape and dv.shape == kd.shape
        ke, ve = (np.ascontiguousarray(np.repeat(x, G, axis=1)) for x in (k, v))
        rq, rk, rv = oracle_mod.attn_bwd_f64(q, ke, ve, do, causal)
        rk, rv = (x.reshape(B, Hkv, G, N, D).sum(2) for x in (rk, rv))
        for name, g, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
            g = g.cpu().numpy()
            assert np.isfinite(g).all(), (name, B, Hq, Hkv, N)
            assert rel(g, ref) < TOL[dtype], (name, causal, B, Hq, Hkv, N, rel(g, ref))
    # through torch.autograd (row f4): grouped heads against torch's own attention on the repeated K / V
    from flash_attention_metal_amd.torch_op import attention_forward

    q, _, _ = make_qkv(oracle_mod, 1, 4, 160, D, dtype)
    _, k, v = make_qkv(oracle_mod, 1, 2, 160, D, dtype)
    qd, kd, vd = (to_dev(x, dtype).requires_grad_(True) for x in (q, k, v))
    o, _ = attention_forward(qd, kd, vd, causal)
    w = torch.linspace(-1, 1, o.numel(), device="cuda").reshape(o.shape)
    (o.float() * w).sum().backward()
    q2, k2, v2 = (to_dev(x, dtype).double().requires_grad_(True) for x in (q, k, v))
    o2 = torch.nn.functional.scaled_dot_product_attention(q2, k2.repeat_interleave(2, 1), v2.repeat_interleave(2, 1), is_causal=causal)
    (o2 * w.to(dtype=getattr(torch, {"bf16": "bfloat16", "f16": "float16"}[dtype])).double()).sum().backward()
    for a, b_ in ((qd.grad, q2.grad), (kd.grad, k2.grad), (vd.grad, v2.grad)):
        assert a.shape == b_.shape
        assert (a.double() - b_).abs().max().item() / b_.abs().max().item() < 3e-2


def rect_reference(q, k, v, do, causal):
    """fp64 gradients of softmax(q k^T / sqrt(D) [+ bottom-right causal mask]) v for q [B,Hq,Nq,D], k / v [B,Hkv,Nk,D]
    (plain numpy: the C oracle's backward is square). dS = P o (dP - rowsum(dO o O)), kernels.metal:1160-1169."""
    B, Hq, Nq, D = q.shape
    Hkv, Nk = k.shape[1], k.shape[2]
    G = Hq // Hkv
    q64, do64 = q.astype(np.float64), do.astype(np.float64)
    ke, ve = (np.repeat(x.astype(np.float64), G, axis=1) for x in (k, v))
    s_ = np.einsum("bhid,bhjd->bhij", q64, ke) / np.sqrt(D)
    if causal:
        i, j = np.arange(Nq)[:, None], np.arange(Nk)[None, :]
        s_ = np.where(j <= i + (Nk - Nq), s_, -np.inf)
    p_ = np.exp(s_ - s_.max(-1, keepdims=True))
    p_ /= p_.sum(-1, keepdims=True)
    o = np.einsum("bhij,bhjd->bhid", p_, ve)
    dv = np.einsum("bhij,bhid->bhjd", p_, do64)
    dp = np.einsum("bhid,bhjd->bhij", do64, ve)
    ds = p_ * (dp - (do64 * o).sum(-1, keepdims=True)) / np.sqrt(D)
    dq = np.einsum("bhij,bhjd->bhid", ds, ke)
    dk = np.einsum("bhij,bhid->bhjd", ds, q64)
    return dq, dk.reshape(B, Hkv, G, Nk, D).sum(2), dv.reshape(B, Hkv, G, Nk, D).sum(2)


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_rectangular(fa, oracle_mod, causal, D):
    """fa_bwd_ex with Nq != Nk (the counterpart of fa_fwd_ex): cross-attention shapes, and bottom-right aligned causal masks."""
    import torch

    dtype = "bf16"
    cases = [(1, 2, 2, 100, 260), (2, 4, 2, 64, 129), (1, 2, 1, 1, 200), (1, 1, 1, 130, 131)]
    if not causal:
        cases += [(1, 2, 2, 260, 100), (1, 4, 1, 200, 1)]  # more queries than keys: non-causal only (fa_fwd_ex's rule)
    for (B, Hq, Hkv, Nq, Nk) in cases:
        q, _, _ = make_qkv(oracle_mod, B, Hq, Nq, D, dtype)
        _, k, v = make_qkv(oracle_mod, B, Hkv, Nk, D, dtype)
        do = oracle_mod.round_to(oracle_mod.init_random(B * Hq * Nq * D, 45).reshape(B, Hq, Nq, D), dtype)
        qd, kd, vd, dod = (to_dev(x, dtype) for x in (q, k, v, do))
        o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)
        dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o, dod, lse, is_causal=causal)
        torch.cuda.synchronize()
        assert dq.shape == qd.shape and dk.shape == kd.shape and dv.shape == kd.shape
        for name, g, ref in zip(("dq", "dk", "dv"), (dq, dk, dv), rect_reference(q, k, v, do, causal)):
            g = g.cpu().numpy()
            assert np.isfinite(g).all(), (name, B, Hq, Hkv, Nq, Nk)
            assert rel(g, ref) < TOL[dtype], (name, causal, B, Hq, Hkv, Nq, Nk, rel(g, ref))
    if causal:
        x = torch.zeros(1, 1, 256, D, dtype=torch.bfloat16, device="cuda")
        y = torch.zeros(1, 1, 128, D, dtype=torch.bfloat16, device="cuda")
        with pytest.raises(fa.FaError) as e:
            fa.flash_attention_backward(x, y, y, x, x, torch.zeros(1, 1, 256, device="cuda"), is_causal=True)
        assert e.value.status == -2


@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
def test_backward_partial_last_key_tile_with_strongly_negative_scores(fa, oracle_mod, D):
    """Non-causal, Nk % 64 != 0, every score strongly negative (k = -8 q direction, f16): lse << 0, so a key slot past Nk -- whose
    K / V rows arrive as zeros -- would give P = exp(-lse) and overflow the cast of dS (inf x 0 = NaN in the whole dQ row) unless the
    dQ kernel masks keys >= Nk in the partial last tile as the forward does."""
    import torch

    dtype = "f16"
    for (B, Hq, Hkv, Nq, Nk) in ((1, 2, 2, 96, 77), (1, 2, 1, 130, 129), (1, 1, 1, 64, 260)):
        rng = np.random.default_rng(Nk)
        u = rng.standard_normal((B, Hkv, 1, D)).astype(np.float32)
        u /= np.sqrt((u ** 2).sum(-1, keepdims=True))
        u *= np.float32((D / 64.0) ** 0.25)  # logits around -25 at every head dim
        q = oracle_mod.round_to(np.repeat(u, Hq // Hkv, axis=1) * 5.0 + 0.05 * rng.standard_normal((B, Hq, Nq, D)).astype(np.float32), dtype)
        k = oracle_mod.round_to(-8.0 * u * 5.0 + 0.05 * rng.standard_normal((B, Hkv, Nk, D)).astype(np.float32), dtype)
        v = oracle_mod.round_to(rng.uniform(-1, 1, (B, Hkv, Nk, D)).astype(np.float32), dtype)
        do = oracle_mod.round_to(rng.uniform(-1, 1, (B, Hq, Nq, D)).astype(np.float32), dtype)
        qd, kd, vd, dod = (to_dev(x, dtype) for x in (q, k, v, do))
        o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=False)
        assert lse.max().item() < -11.2  # the regime the advisory describes: exp(-lse) is beyond the f16 range
        dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o, dod, lse, is_causal=False)
        torch.cuda.synchronize()
        for name, g, ref in zip(("dq", "dk", "dv"), (dq, dk, dv), rect_reference(q, k, v, do, False)):
            g = g.cpu().numpy()
            assert np.isfinite(g).all(), (name, Nq, Nk, D)
            # (logits around -25: the rounding of the pre-scaled operands moves every score by ~1e-2, section "LSE accuracy" of the header;
            # the point here is a finite, sane gradient -- before the fix the whole dQ row was NaN)
            # (head_dim 32: dS = P (dP - delta) cancels over fewer columns, measured 0.105 on dQ)
            assert rel(g, ref) < (0.1 if D >= 64 else 0.2), (name, Nq, Nk, D, rel(g, ref))


def test_backward_known_answers(fa, oracle_mod):
    import torch

    # causal, row 0 attends to key 0 only: P = 1 -> dS = 0 -> dQ[0] = 0 exactly; dV[N-1] = P[N-1,N-1] * dO[N-1]
    N = 192
    q, k, v = make_qkv(oracle_mod, 1, 2, N, 64, "bf16")
    do = oracle_mod.round_to(oracle_mod.init_random(2 * N * 64, 45).reshape(1, 2, N, 64), "bf16")
    dq, dk, dv = grads(fa, q, k, v, do, "bf16", True)
    # (zero up to the fp32 rounding of dP - delta: delta enters the dP chain as its initial accumulator, so the two equal sums
    #  are formed in different orders; round 2 subtracted them after the fact and happened to cancel exactly)
    assert np.abs(dq[:, :, 0]).max() < 1e-6
    # V = const: O = const, dP_ij = dO_i . v is the same for every j -> dS = 0 -> dQ = dK = 0 up to rounding
    vc = np.full_like(v, 0.5)
    dq, dk, dv = grads(fa, q, k, vc, do, "bf16", False)
    assert np.abs(dq).max() < 2e-3 and np.abs(dk).max() < 2e-3
    # column sums: sum_j dV[j,:] = sum_i dO[i,:] (softmax rows sum to 1)
    np.testing.assert_allclose(dv.sum(2), do.sum(2), rtol=0, atol=2e-2 * np.abs(do.sum(2)).max() + 0.05)


def test_backward_deterministic_and_matches_autograd(fa, oracle_mod):
    import torch
    import torch.nn.functional as F

    q, k, v = make_qkv(oracle_mod, 2, 2, 384, 64, "bf16")
    do = oracle_mod.round_to(oracle_mod.init_random(q.size, 45).reshape(q.shape), "bf16")
    a = grads(fa, q, k, v, do, "bf16", True)
    b = grads(fa, q, k, v, do, "bf16", True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)  # no atomics: bitwise reproducible
    tq, tk, tv = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v))
    F.scaled_dot_product_attention(tq, tk, tv, is_causal=True).backward(torch.tensor(do, dtype=torch.float64))
    for g, t in zip(a, (tq, tk, tv)):
        assert rel(g, t.grad.numpy()) < TOL["bf16"]


def test_backward_config4_shape_sampled_head(fa, oracle_mod):
    # BASELINE config 4 shape family (head_dim 128, causal) at reduced size: one head against the fp64 oracle
    import torch

    B, H, N, D = 1, 2, 2048, 128
    g = torch.Generator(device="cuda").manual_seed(6)
    q, k, v, do = (torch.rand(B, H, N, D, generator=g, device="cuda").mul_(2).sub_(1).to(torch.bfloat16) for _ in range(4))
    o, lse = fa.flash_attention_forward(q, k, v, is_causal=True)
    dq, dk, dv = fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=True)
    torch.cuda.synchronize()
    f = lambda t: np.ascontiguousarray(t[:, 1:2].float().cpu().numpy())  # noqa: E731
    rq, rk, rv = oracle_mod.attn_bwd_f64(f(q), f(k), f(v), f(do), True)
    for gq, ref in ((dq, rq), (dk, rk), (dv, rv)):
        assert torch.isfinite(gq).all() and rel(gq[:, 1:2].cpu().numpy(), ref) < TOL["bf16"]


def test_backward_config3_shape_sampled_head(fa, oracle_mod):
    # BASELINE config 3 shape family at reduced batch (B=1,H=4,N=4096): one head against the fp64 oracle
    import torch

    B, H, N = 1, 4, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    q, k, v, do = (torch.rand(B, H, N, 64, generator=g, device="cuda").mul_(2).sub_(1).to(torch.bfloat16) for _ in range(4))
    o, lse = fa.flash_attention_forward(q, k, v, is_causal=True)
    dq, dk, dv = fa.flash_attention_backward(q, k, v, o, do, lse, is_causal=True)
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in (dq, dk, dv))
    h = 3
    f = lambda t: np.ascontiguousarray(t[:, h:h + 1].float().cpu().numpy())  # noqa: E731
    rq, rk, rv = oracle_mod.attn_bwd_f64(f(q), f(k), f(v), f(do), True)
    for gq, ref in ((dq, rq), (dk, rk), (dv, rv)):
        assert rel(gq[:, h:h + 1].cpu().numpy(), ref) < TOL["bf16"]


def test_backward_errors(fa):
    import torch

    lse = torch.zeros(1, 1, 128, device="cuda")
    for D in (36, 136, 512):  # the backward covers multiples of 8 up to 128, and 256
        x = torch.zeros(1, 1, 128, D, dtype=torch.bfloat16, device="cuda")
        with pytest.raises(fa.FaError) as e:
            fa.flash_attention_backward(x, x, x, x, x, lse)
        assert e.value.status == -2  # reported, not faked


@pytest.mark.parametrize("D", [64, 128, 32, 96])
@pytest.mark.parametrize("causal", [False, True])
def test_backward_fp8_inputs(fa, oracle_mod, causal, D):
    """e4m3 Q, K, V (BASELINE config 5's family) with the bf16 O the forward writes for them and a bf16 dO: fa_bwd_ex widens Q, K, V to
    bf16 -- exactly -- into its workspace (fa_bwd_workspace_bytes_ex) and runs the bf16 kernels, so the bf16 bar applies against the
    fp64 oracle on the e4m3 values. O and LSE come from the bf16-probability forward (variant mfma_exact: the all-fp8 forward AUTO takes
    on large grids rounds its probabilities to e4m3 and carries 2^-4 on O and LSE, include/fa_mi355.h). Grouped heads and Nq != Nk too."""
    import torch

    for (B, Hq, Hkv, Nq, Nk) in ((1, 2, 2, 200, 200), (2, 4, 2, 129, 129), (1, 2, 1, 100, 260)):
        q, _, _ = make_qkv(oracle_mod, B, Hq, Nq, D, "fp8", amp=2.0)
        _, k, v = make_qkv(oracle_mod, B, Hkv, Nk, D, "fp8", amp=2.0)
        do = oracle_mod.round_to(oracle_mod.init_random(B * Hq * Nq * D, 45).reshape(B, Hq, Nq, D), "bf16")
        qd, kd, vd = (to_dev(x, "fp8") for x in (q, k, v))
        dod = to_dev(do, "bf16")
        if D in (64, 128):
            o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal, variant="mfma_exact")
        else:  # the forward has no e4m3 kernel for these head dims: O and LSE of the same values through the bf16 kernel (e4m3 -> bf16 is exact)
            o, lse = fa.flash_attention_forward(to_dev(q, "bf16"), to_dev(k, "bf16"), to_dev(v, "bf16"), is_causal=causal, variant="mfma_exact")
        assert o.dtype == torch.bfloat16
        dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o, dod, lse, is_causal=causal)
        torch.cuda.synchronize()
        assert dq.shape == qd.shape and dk.shape == kd.shape and dq.dtype == torch.float32
        for name, g, ref in zip(("dq", "dk", "dv"), (dq, dk, dv), rect_reference(q, k, v, do, causal)):
            g = g.cpu().numpy()
            assert np.isfinite(g).all(), (name, B, Hq, Hkv, Nq, Nk)
            assert rel(g, ref) < TOL["bf16"], (name, causal, D, B, Hq, Hkv, Nq, Nk, rel(g, ref))
    # bit-identical to the bf16 backward on the widened tensors (the widening is exact)
    q, k, v = make_qkv(oracle_mod, 1, 2, 192, D, "fp8")
    do = oracle_mod.round_to(oracle_mod.init_random(2 * 192 * D, 46).reshape(1, 2, 192, D), "bf16")
    qb, kb, vb, dod = (to_dev(x, "bf16") for x in (q, k, v, do))
    o, lse = fa.flash_attention_forward(qb, kb, vb, is_causal=causal)
    g8 = fa.flash_attention_backward(to_dev(q, "fp8"), to_dev(k, "fp8"), to_dev(v, "fp8"), o, dod, lse, is_causal=causal)
    gb = fa.flash_attention_backward(qb, kb, vb, o, dod, lse, is_causal=causal)
    for a_, b_ in zip(g8, gb):
        assert torch.equal(a_, b_)
    # through torch.autograd (row f4): e4m3 leaves, bf16 O; the gradients come back in the leaves' dtype
    import flash_attention_metal_amd.torch_op  # noqa: F401  (registers the op)

    if D in (64, 128):
        q8, k8, v8 = (to_dev(x, "fp8").requires_grad_(True) for x in (q, k, v))
        o8, _ = torch.ops.fa_mi355.attention_forward(q8, k8, v8, causal, 0.0)
        (o8.float() * dod.float()).sum().backward()
        assert q8.grad is not None and q8.grad.dtype == q8.dtype and k8.grad.shape == k8.shape
        o_ref, lse_ref = fa.flash_attention_forward(q8.detach(), k8.detach(), v8.detach(), is_causal=causal)
        gq, gk, gv = fa.flash_attention_backward(q8.detach(), k8.detach(), v8.detach(), o_ref, dod, lse_ref, is_causal=causal)
        for a_, b_ in zip((q8.grad, k8.grad, v8.grad), (gq, gk, gv)):
            assert torch.equal(a_.float(), b_.to(a_.dtype).float())


@pytest.mark.parametrize("D", [8, 16, 40, 48, 80, 112, 120])
def test_backward_any_multiple_of_eight(fa, oracle_mod, D):
    """Head dims other than 64 / 128 run the next larger kernel on zero-padded rows (fa_bwd_kernels.hip, PAD): any multiple of 8 up
    to 128, causal and full, a ragged length with a partial last tile on both sides. The forward that supplies O and LSE here is the
    fp64 oracle's (the forward kernels cover 32 / 64 / 96 / 128 / 256 only)."""
    import torch

    dtype = "bf16"
    B, H, N = 1, 2, 203
    q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
    do = oracle_mod.round_to(oracle_mod.init_random(B * H * N * D, 45).reshape(B, H, N, D), dtype)
    for causal in (False, True):
        o64, lse64 = oracle_mod.attn_fwd_f64(q, k, v, causal)
        qd, kd, vd, dod = (to_dev(x, dtype) for x in (q, k, v, do))
        od = to_dev(oracle_mod.round_to(o64.astype(np.float32), dtype), dtype)
        lsed = torch.from_numpy(lse64.astype(np.float32)).cuda()
        dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, od, dod, lsed, is_causal=causal)
        torch.cuda.synchronize()
        for name, g, ref in zip(("dq", "dk", "dv"), (dq, dk, dv), oracle_mod.attn_bwd_f64(q, k, v, do, causal)):
            g = g.cpu().numpy()
            assert np.isfinite(g).all(), (name, D, causal)
            assert rel(g, ref) < TOL[dtype], (name, D, causal, rel(g, ref))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("causal", [False, True])
def test_torch_op_autograd_matches_sdpa(fa, oracle_mod, dtype, causal):
    # fa_bwd is reachable from torch.autograd through the custom op. Gradients of a random linear
    # functional of O, against scaled_dot_product_attention differentiated in fp64 on the same (rounded) inputs.
    import torch
    import torch.nn.functional as F

    from flash_attention_metal_amd import torch_op  # noqa: F401  (registers the op and its autograd formula)
    from util import make_qkv, to_dev

    B, H, N, D = 2, 3, 200, 64
    q, k, v = (to_dev(x, dtype).requires_grad_(True) for x in make_qkv(oracle_mod, B, H, N, D, dtype))
    w = torch.randn(B, H, N, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    o, lse = torch.ops.fa_mi355.attention_forward(q, k, v, causal, 0.0)
    (o.float() * w).sum().backward()
    q64, k64, v64 = (x.detach().double().requires_grad_(True) for x in (q, k, v))
    ref = F.scaled_dot_product_attention(q64, k64, v64, is_causal=causal)
    (ref * w.double()).sum().backward()
    tol = {"f16": 4e-3, "bf16": 2e-2}[dtype]
    for g, g64, name in ((q.grad, q64.grad, "dq"), (k.grad, k64.grad, "dk"), (v.grad, v64.grad, "dv")):
        assert g is not None and g.dtype == q.dtype
        err = (g.double() - g64).abs().max().item()
        assert err < tol * g64.abs().max().item(), (name, err, g64.abs().max().item())
    # shapes without a backward kernel raise instead of handing back a silent zero gradient
    q32 = torch.zeros(1, 1, 64, 64, dtype=torch.float32, device="cuda", requires_grad=True)  # fp32 inputs: a forward (scalar kernels), no backward
    o32, _ = torch.ops.fa_mi355.attention_forward(q32, q32.detach(), q32.detach(), False, 0.0)
    with pytest.raises(Exception):
        o32.float().sum().backward()
    # head_dim 96 (zero-padded rows of the 128 kernel) through autograd
    q96, k96, v96 = (to_dev(x, dtype).requires_grad_(True) for x in make_qkv(oracle_mod, 1, 2, 130, 96, dtype))
    o96, _ = torch.ops.fa_mi355.attention_forward(q96, k96, v96, causal, 0.0)
    w96 = torch.randn(1, 2, 130, 96, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    (o96.float() * w96).sum().backward()
    r96 = [x.detach().double().requires_grad_(True) for x in (q96, k96, v96)]
    (F.scaled_dot_product_attention(*r96, is_causal=causal) * w96.double()).sum().backward()
    for g, g64 in zip((q96.grad, k96.grad, v96.grad), (x.grad for x in r96)):
        assert (g.double() - g64).abs().max().item() < tol * g64.abs().max().item()
    o2, lse2 = torch.ops.fa_mi355.attention_forward(q, k, v, causal, 0.0)
    with pytest.raises(Exception):  # no gradient through the LSE output
        lse2.sum().backward()
    torch.cuda.synchronize()


@pytest.mark.parametrize("D", [64, 128])
def test_backward_padded_strides_and_larger_logits(fa, oracle_mod, D):
    """Tensors that are views into padded buffers (head stride > N*D, batch stride > H*head stride; Q and K/V padded differently)
    and inputs three times larger than the benchmark's (|score| up to ~20): the strides reach the kernels through fa_bwd_ex, the
    row constants -LSE*log2e ride in the accumulators at that magnitude."""
    import torch

    dtype, causal = "bf16", True
    B, Hq, Hkv, N = 2, 4, 2, 200
    q, _, _ = make_qkv(oracle_mod, B, Hq, N, D, dtype)
    _, k, v = make_qkv(oracle_mod, B, Hkv, N, D, dtype)
    q, k = (oracle_mod.round_to(x * 3.0, dtype) for x in (q, k))
    do = oracle_mod.round_to(oracle_mod.init_random(B * Hq * N * D, 45).reshape(B, Hq, N, D), dtype)

    def padded(x, extra_rows, extra_heads):
        b, h, n, d = x.shape
        buf = torch.full((b, h + extra_heads, n + extra_rows, d), float("nan"), dtype=torch.bfloat16, device="cuda")
        view = buf[:, :h, :n]
        view.copy_(to_dev(x, dtype))
        return view

    qd, dod = padded(q, 8, 1), padded(do, 8, 1)
    kd, vd = padded(k, 24, 0), padded(v, 24, 0)
    assert not qd.is_contiguous() and qd.stride() == dod.stride() and kd.stride() == vd.stride() and qd.stride(1) != kd.stride(1)
    o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)
    o_p = padded(o.float().cpu().numpy(), 8, 1)  # O in a buffer padded like Q's (the backward wants q, o, d_o under one stride pair)
    dq, dk, dv = fa.flash_attention_backward(qd, kd, vd, o_p, dod, lse, is_causal=causal)
    torch.cuda.synchronize()
    assert dq.stride() == qd.stride() and dk.stride() == kd.stride()
    ke, ve = (np.ascontiguousarray(np.repeat(x, Hq // Hkv, axis=1)) for x in (k, v))
    rq, rk, rv = oracle_mod.attn_bwd_f64(q, ke, ve, do, causal)
    rk, rv = (x.reshape(B, Hkv, Hq // Hkv, N, D).sum(2) for x in (rk, rv))
    for name, g, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        g = g.cpu().numpy()
        assert np.isfinite(g).all(), name  # (the NaN padding was never read)
        assert rel(g, ref) < TOL[dtype], (name, D, rel(g, ref))
