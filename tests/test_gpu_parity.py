"""GPU parity: every kernel, through the C-ABI, against the CPU oracle.

Oracle = oracle/ (C restatement of /root/reference/main.mm:128-159, :551-578,
pinned bit-exact to the reference's own loops by tests/test_oracle.py).
Tolerances (max-abs on O, inputs U(-1,1) so |O| <= 1):
  fp32 kernels  : 2e-5   (the reference accepts 1e-3, main.mm:239,253,292)
  fp16 MFMA     : 1.5e-3 (reference: 5e-3 main.mm:375 / 1e-2 main.mm:452)
  bf16 MFMA     : 6e-3   (bf16 has 3 fewer mantissa bits than fp16; reference bar for
                          its 16-bit operator is 1e-2, main.mm:452,591)
  LSE           : 2e-5 fp32 kernels, 1e-4 MFMA (fp32 accumulate; unpinned in the reference)
The kernels that pre-scale the query operand (variants mfma, mfma_split2 for f16/bf16, D <= 128) compute the exact
operator on Q~ = round(scale*log2e*Q): they are held to the SAME tolerances against the oracle evaluated on that Q~
(util.effective_q reproduces it bit for bit), and to the bound include/fa_mi355.h states ("LSE accuracy") against the
oracle on the true Q. Variant mfma_exact is the same kernel without the pre-scaling.
Index logic (causal mask, tile skip, head/batch addressing) is checked bit-exact.
"""
import numpy as np
import pytest

from util import LN2, effective_q, fp8pv_lse_term, fp8pv_term, is_prescaled, lse_tol, make_qkv, o_tol, rowsum_term, run_op, to_dev

pytestmark = pytest.mark.gpu

TOL_O = {"f32": 2e-5, "f16": 1.5e-3, "bf16": 6e-3}
TOL_LSE = {"f32": 2e-5, "f16": 1e-4, "bf16": 1e-4}
# the matrix-core kernels by name; "auto" picks by grid size
MFMA_VARIANTS = ["mfma", "mfma_splitkv", "mfma_split2", "mfma_exact", "mfma_h64s2", "mfma16", "mfma_fp8pv"]


def need(fa, dtype, variant, D):
    if not fa.supported({"fp8": "fp8_e4m3"}.get(dtype, dtype), variant, D):
        pytest.skip(f"{variant} has no kernel for {dtype} D={D}")


@pytest.fixture(scope="module")
def fa():
    import torch

    import flash_attention_metal_amd as fa

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    fa.load_library()  # raises if the HIP library is missing: no silent fallback
    return fa


def check(fa, oracle, q, k, v, dtype, causal, variant, tol_scale=1.0, scale=None):
    o, lse = run_op(fa, q, k, v, dtype, causal, variant, scale)
    assert np.isfinite(o).all() and np.isfinite(lse).all()
    pre = is_prescaled(fa, dtype, variant, *q.shape, causal)
    if pre:  # strict: the exact operator on the operand the kernel really multiplies
        o64, lse64 = oracle.attn_fwd_f64(effective_q(oracle, q, dtype, scale), k, v, causal, LN2)
        err_o, err_l = np.abs(o - o64).max(), np.abs(lse - lse64).max()
        assert err_o < TOL_O[dtype] * tol_scale, (variant, dtype, causal, q.shape, err_o, "vs oracle on Q~")
        assert err_l < TOL_LSE[dtype] * tol_scale + rowsum_term(dtype, pre), (variant, dtype, causal, q.shape, err_l, "vs oracle on Q~")
    # against the true Q: the plain tolerances, plus the documented operand-rounding bound where it applies
    o64, lse64 = oracle.attn_fwd_f64(q, k, v, causal, scale)
    err_o = np.abs(o - o64).max()
    err_l = np.abs(lse - lse64).max()
    assert err_o < o_tol(dtype, pre, q, k, v, scale, TOL_O[dtype] * tol_scale), (variant, dtype, causal, q.shape, err_o)
    assert err_l < lse_tol(dtype, pre, q, k, scale, TOL_LSE[dtype] * tol_scale), (variant, dtype, causal, q.shape, err_l)
    return err_o, err_l


# --------------------------------------------------------------------------
# fp32 kernels vs the reference's own outputs (golden) -- main.mm:231-296
# --------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["naive", "tiled", "tiled_v2"])
def test_fp32_variants_vs_reference_golden(fa, oracle_mod, golden, variant):
    g, _ = golden
    D = 64
    for n in (128, 256):  # reference mode: Q == K == V, seed 42
        x = oracle_mod.init_random(n * D, 42).reshape(1, 1, n, D)
        for causal, key in ((False, f"noncausal_same_n{n}"), (True, f"causal_same_n{n}")):
            o, _ = run_op(fa, x, x, x, "f32", causal, variant)
            assert np.abs(o[0, 0] - g[key]).max() < 2e-5, (variant, key)
    for n in (128, 200):  # independent Q,K,V and a ragged N
        q, k, v = [oracle_mod.init_random(n * D, s).reshape(1, 1, n, D) for s in (42, 43, 44)]
        for causal, key in ((False, f"noncausal_indep_n{n}"), (True, f"causal_indep_n{n}")):
            o, _ = run_op(fa, q, k, v, "f32", causal, variant)
            assert np.abs(o[0, 0] - g[key]).max() < 2e-5, (variant, key)


def test_naive_vs_cpu_n1024_the_reference_check(fa, oracle_mod, golden):
    # main.mm:231-242: naive kernel vs CPU oracle at N=1024, Q=K=V, tolerance 1e-3 there
    g, _ = golden
    x = oracle_mod.init_random(1024 * 64, 42).reshape(1, 1, 1024, 64)
    for variant in ("naive", "tiled", "tiled_v2"):
        o, _ = run_op(fa, x, x, x, "f32", False, variant)
        assert np.abs(o[0, 0][::16] - g["noncausal_same_n1024_rows_step16"]).max() < 2e-5


@pytest.mark.parametrize("variant", ["naive", "tiled", "tiled_v2"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_scalar_variants_batched_ragged(fa, oracle_mod, variant, dtype):
    for (B, H, N, D) in ((2, 3, 77, 64), (1, 2, 130, 128), (1, 1, 65, 32)):
        q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
        for causal in (False, True):
            check(fa, oracle_mod, q, k, v, dtype, causal, variant, tol_scale=3.0 if dtype != "f32" else 1.0)


# --------------------------------------------------------------------------
# the matrix-core operator
# --------------------------------------------------------------------------
@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_mfma_vs_oracle(fa, oracle_mod, dtype, D, causal, variant):
    need(fa, dtype, variant, D)
    # N: multiples of the tile, ragged, < one tile, > several q blocks
    for (B, H, N) in ((1, 1, 128), (2, 3, 200), (1, 2, 1), (1, 1, 63), (1, 2, 65), (1, 1, 129), (2, 2, 1000),
                      (1, 8, 1024), (1, 2, 255), (1, 1, 257), (1, 3, 576)):
        q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
        check(fa, oracle_mod, q, k, v, dtype, causal, variant)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_mfma_fp8_inputs_vs_oracle(fa, oracle_mod, D, causal, variant):
    need(fa, "fp8", variant, D)
    # BASELINE config 5 family: Q,K,V OCP e4m3fn (saturating RNE), fp32 accumulate, bf16 O.
    # The oracle sees exactly the e4m3 values, so only P/O rounding (bf16) separates the two.
    for (B, H, N) in ((1, 1, 128), (2, 3, 200), (1, 2, 65), (2, 2, 1000)):
        for amp in (1.0, 3.0):  # amp 3: values up to 3 exercise more of the e4m3 grid
            q, k, v = make_qkv(oracle_mod, B, H, N, D, "fp8", amp=amp)
            o, lse = run_op(fa, q, k, v, "fp8", causal, variant)
            o64, l64 = oracle_mod.attn_fwd_f64(q, k, v, causal)
            assert np.abs(o - o64).max() < TOL_O["bf16"] * amp + fp8pv_term(variant, "fp8", v), (B, H, N, D, causal, amp, np.abs(o - o64).max())
            assert np.abs(lse - l64).max() < 1e-4 * amp * amp + fp8pv_lse_term(variant, "fp8")


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
def test_fp8_equals_bf16_kernel_on_same_values(fa, oracle_mod, variant):
    # e4m3 -> bf16 is exact, so the fp8-input kernel must reproduce the bf16 kernel bit for bit
    import torch

    need(fa, "fp8", variant, 64)
    q, k, v = make_qkv(oracle_mod, 2, 4, 320, 64, "fp8", amp=2.0)
    for causal in (False, True):
        o8, l8 = run_op(fa, q, k, v, "fp8", causal, variant)
        if variant != "mfma_fp8pv":
            ob, lb = run_op(fa, q, k, v, "bf16", causal, variant)
        if variant == "mfma_fp8pv":  # probabilities rounded to e4m3: close to the bf16 kernel, not equal
            ob, lb = run_op(fa, q, k, v, "bf16", causal, "mfma_exact")
            assert np.abs(l8 - lb).max() < 5e-5 + fp8pv_lse_term(variant, "fp8")
            assert np.abs(o8 - ob).max() <= fp8pv_term(variant, "fp8", v) and np.sqrt(((o8 - ob) ** 2).mean()) < 0.1 * fp8pv_term(variant, "fp8", v)
        elif variant in ("mfma", "mfma_split2", "mfma_exact"):
            # these kernels (one body) multiply e4m3 by e4m3 on the scaled fp8 MFMA (64 head-dim elements per instruction): every
            # product is exact, but fp32 partial sums are formed in a different order than in the bf16 instruction, so
            # the scores agree to fp32 rounding, not bit for bit
            # (and the bf16 kernel pre-scales its query operand, the fp8 score product cannot: compare with mfma_exact)
            ob, lb = run_op(fa, q, k, v, "bf16", causal, "mfma_exact")
            assert np.abs(l8 - lb).max() < 5e-5  # |lse| <= 7 here; the oracle tolerance for LSE is 1e-4
            assert np.abs(o8 - ob).max() <= 2 ** -7 * np.abs(ob).max()  # a bf16 ulp or two on a few elements
        else:  # e4m3 widened to bf16 while staging: the very same arithmetic
            assert np.array_equal(o8, ob) and np.array_equal(l8, lb)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
def test_mfma_reference_mode_same_qkv_n1024(fa, oracle_mod, variant):
    need(fa, "f16", variant, 64)
    # main.mm:381-456 (V4 vs naive, N=1024, Q=K=V, fp16, tol 1e-2) and :458-594 (causal N=128)
    x = oracle_mod.round_to(oracle_mod.init_random(1024 * 64, 42).reshape(1, 1, 1024, 64), "f16")
    check(fa, oracle_mod, x, x, x, "f16", False, variant)
    xc = x[:, :, :128].copy()
    check(fa, oracle_mod, xc, xc, xc, "f16", True, variant)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_causal_row0_is_v0_bit_exact(fa, oracle_mod, dtype, variant):
    need(fa, dtype, variant, 64)
    q, k, v = make_qkv(oracle_mod, 2, 2, 300, 64, dtype)
    o, lse = run_op(fa, q, k, v, dtype, True, variant)
    assert np.array_equal(o[:, :, 0], v[:, :, 0])  # softmax over one key: O[0] == V[0]


@pytest.mark.parametrize("variant,dtype", [("mfma", "bf16"), ("mfma", "f16"), 
                                           ("mfma_splitkv", "bf16"), ("mfma_splitkv", "f16"), ("mfma_split2", "bf16"), ("mfma_split2", "f16"),
                                           ("mfma_exact", "bf16"), ("mfma_h64s2", "bf16"), ("mfma_h64s2", "f16"), ("mfma16", "bf16"), ("mfma16", "f16"), ("mfma", "fp8"), ("mfma_splitkv", "fp8"), ("mfma_split2", "fp8"), ("mfma_fp8pv", "fp8"),
                                           ("tiled_v2", "f32"), ("tiled", "f32"), ("naive", "f32")])
def test_mask_index_probe_exact(fa, oracle_mod, variant, dtype):
    # Q = 0 -> uniform softmax; V[j,0] = delta(j,t): causal O[i,0] = 1/(i+1) for i >= t, EXACTLY 0 left of it.
    # t straddles every tile / wave / block boundary of the kernels (32, 64, 128).
    N, D = 320, 64
    q = np.zeros((1, 1, N, D), np.float32)
    k = make_qkv(oracle_mod, 1, 1, N, D, dtype)[1]
    for t in (0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 319):
        v = np.zeros((1, 1, N, D), np.float32)
        v[0, 0, t, 0] = 1.0
        o, lse = run_op(fa, q, k, v, dtype, True, variant)
        col = o[0, 0, :, 0]
        assert np.array_equal(col[:t], np.zeros(t, np.float32)), (variant, t)
        i = np.arange(N, dtype=np.float32)
        expect = oracle_mod.round_to(np.float32(1.0) / (i + 1), "bf16" if dtype == "fp8" else dtype)  # fp8 inputs: O is bf16
        ulp = {"f32": 2e-7, "f16": 1e-3, "bf16": 8e-3, "fp8": 8e-3}[dtype]
        assert np.all(np.abs(col[t:] - expect[t:]) <= ulp * expect[t:]), (variant, t)
        assert np.count_nonzero(o[0, 0, :, 1:]) == 0
        assert np.abs(lse[0, 0] - np.log(i + 1)).max() < 1e-5
        # non-causal: every row sees key t
        o, _ = run_op(fa, q, k, v, dtype, False, variant)
        assert np.all(np.abs(o[0, 0, :, 0] - oracle_mod.round_to(np.float32(1.0 / N), "bf16" if dtype == "fp8" else dtype)) <= ulp / N)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_forced_rescale_branch(fa, oracle_mod, dtype, variant):
    need(fa, dtype, variant, 64)
    # cdna guide rule 26: force the running max to jump at chosen tiles. Key j* is a spiked copy of
    # query i*, so row i* meets a much larger score at tile j*/64 (and the wave takes its rescale path).
    B, H, N, D = 1, 2, 512, 64
    q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
    for (istar, jstar) in ((5, 130), (300, 3), (300, 290), (511, 448), (64, 64), (200, 199)):
        k[:, :, jstar] = oracle_mod.round_to(q[:, :, istar] * 6.0, dtype)
    for causal in (False, True):
        check(fa, oracle_mod, q, k, v, dtype, causal, variant)
    # large-magnitude scores: exercises exp2 range and max tracking (scale folded in log2 domain)
    q2, k2, v2 = make_qkv(oracle_mod, 1, 1, 256, 64, dtype, amp=4.0)
    check(fa, oracle_mod, q2, k2, v2, dtype, True, variant, tol_scale=4.0)
    # deferred-max kernels: the row max creeps up tile after tile by less than the rescale
    # threshold (2^8), then jumps far above it: exercises both the deferred and the taken path, on every
    # row of one block and on a single row of another
    N2 = 1024
    q3, k3, v3 = make_qkv(oracle_mod, 1, 2, N2, D, dtype, seeds=(7, 8, 9))
    ramp = (np.arange(N2, dtype=np.float32) / N2)[None, None, :, None]  # keys grow towards the end
    k3 = oracle_mod.round_to(k3 * 0.25 + q3[:, :, 100:101] * ramp * 3.0, dtype)   # scores of row 100 ramp up smoothly
    k3[:, :, 900] = oracle_mod.round_to(q3[:, :, 100] * 8.0, dtype)              # ... and jump at key 900
    for causal in (False, True):
        check(fa, oracle_mod, q3, k3, v3, dtype, causal, variant, tol_scale=2.0)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("D", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_reference_max_overflow_path(fa, oracle_mod, dtype, D, variant):
    # The 128-row kernel forms P with a stale reference max and lets the ROW SUMS say when that max is too old: a
    # score so far above it that exp2 overflows to +inf in fp32 must send the wave down the exact path (scores
    # recomputed from LDS, max, rescale). Spikes of every size -- below the 2^8 threshold, above it, past the fp32
    # exponent range -- at tile starts, inside masked diagonal tiles, in the last (ragged) tile, on one row and on all rows.
    need(fa, dtype, variant, D)
    B, H, N = 1, 2, 600
    q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype, seeds=(11, 12, 13))
    qn = (q * q).sum(-1, keepdims=True)  # |q_i|^2: a key c*q_i/|q_i|^2 scores exactly c against row i
    for (i, j, c) in ((40, 3, 30.0), (40, 70, 400.0), (41, 41, 900.0), (200, 130, 25.0), (333, 320, 2000.0),
                      (599, 576, 700.0), (599, 598, 1500.0), (128, 128, 1200.0), (450, 64, 60.0), (450, 449, 3000.0)):
        k[:, :, j] = oracle_mod.round_to(q[:, :, i] / qn[:, :, i] * c, dtype)
    k[:, 1, 500] = oracle_mod.round_to(np.full(D, 6.0, np.float32), dtype)  # one key that lifts MANY rows at once
    q[:, 1, 520:] = oracle_mod.round_to(np.abs(q[:, 1, 520:]) + 2.0, dtype)
    for causal in (False, True):
        check(fa, oracle_mod, q, k, v, dtype, causal, variant, tol_scale=2.0)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_strongly_negative_scores_from_the_first_tile_on(fa, oracle_mod, dtype, variant):
    # The 16x16x32 kernel forms the FIRST tile's probabilities against an assumed row maximum of 0 and checks afterwards that no row's
    # sum vanished (csrc/fa_mfma16_kernel.hip, tile()): rows whose every score is far below zero -- here around -40 ... -1400 in log2
    # units, in all tiles or in the first tile only -- must come out like any other (every kernel runs the case).
    need(fa, dtype, variant, 64)
    B, H, N, D = 1, 3, 300, 64
    rng = np.random.default_rng(77)
    u = rng.standard_normal((1, 1, 1, D)).astype(np.float32)
    u /= np.sqrt((u ** 2).sum())
    for (alpha, beta, first_only) in ((4.0, 60.0, False), (12.0, 60.0, False), (30.0, 250.0, False), (12.0, 60.0, True)):
        q = oracle_mod.round_to(alpha * u + 0.05 * rng.standard_normal((B, H, N, D)).astype(np.float32), dtype)
        k = oracle_mod.round_to(-beta * u + 0.05 * rng.standard_normal((B, H, N, D)).astype(np.float32), dtype)
        if first_only:  # keys 64.. are ordinary: the reference has to climb by hundreds of units at the second tile
            k[:, :, 64:] = oracle_mod.round_to(rng.uniform(-1, 1, (B, H, N - 64, D)).astype(np.float32), dtype)
        v = make_qkv(oracle_mod, B, H, N, D, dtype, seeds=(3, 4, 5))[2]
        for causal in (False, True):
            check(fa, oracle_mod, q, k, v, dtype, causal, variant, tol_scale=2.0)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("D", [64, 128, 256])
def test_reference_max_overflow_path_fp8(fa, oracle_mod, D, variant):
    # the same through the e4m3 score product: a few keys at the top of the e4m3 range (448) against rows of ~2
    need(fa, "fp8", variant, D)
    B, H, N = 1, 2, 600
    q, k, v = make_qkv(oracle_mod, B, H, N, D, "fp8", seeds=(11, 12, 13), amp=2.0)
    for j, c in ((3, 16.0), (70, 64.0), (130, 448.0), (320, 208.0), (576, 448.0), (598, 320.0)):
        k[:, :, j] = oracle_mod.round_to(np.sign(q[:, :, (j * 7) % N]) * c, "fp8")
    for causal in (False, True):
        o, lse = run_op(fa, q, k, v, "fp8", causal, variant)
        o64, l64 = oracle_mod.attn_fwd_f64(q, k, v, causal)
        assert np.isfinite(o).all() and np.isfinite(lse).all()
        assert np.abs(o - o64).max() < TOL_O["bf16"] * 2.0 + fp8pv_term(variant, "fp8", v), (D, causal, np.abs(o - o64).max())
        assert (np.abs(lse - l64) / np.maximum(1.0, np.abs(l64))).max() < 2e-5 + fp8pv_lse_term(variant, "fp8"), (D, causal)  # = the 1e-4 absolute bar at LSE ~ 5


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
def test_asymmetric_structure(fa, oracle_mod, variant):
    need(fa, "bf16", variant, 64)
    # catches K<->V swaps, transposed S, wrong-row V gathers that Q==K==V data cannot
    N, D = 256, 64
    q, k, _ = make_qkv(oracle_mod, 1, 1, N, D, "bf16")
    v = np.zeros((1, 1, N, D), np.float32)
    v[0, 0] = oracle_mod.round_to((np.arange(N)[:, None] % 7 - 3) * 0.25 + (np.arange(D)[None, :] % 5) * 0.125, "bf16")
    for causal in (False, True):
        check(fa, oracle_mod, q, k, v, "bf16", causal, variant)


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_head_batch_addressing_bit_exact(fa, oracle_mod, dtype, variant):
    need(fa, dtype, variant, 64)
    import functools

    import torch

    B, H, N, D = 3, 5, 320, 64
    q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
    qd, kd, vd = (to_dev(x, dtype) for x in (q, k, v))
    fwd = functools.partial(fa.flash_attention_forward, variant=variant)
    o_all, l_all = fwd(qd, kd, vd, is_causal=True)
    # (1) each (b,h) slice alone gives the same bits as inside the batch
    for b, h in ((0, 0), (1, 3), (2, 4)):
        o1, l1 = fwd(qd[b:b + 1, h:h + 1].contiguous(), kd[b:b + 1, h:h + 1].contiguous(),
                                            vd[b:b + 1, h:h + 1].contiguous(), is_causal=True)
        assert torch.equal(o1[0, 0], o_all[b, h]) and torch.equal(l1[0, 0], l_all[b, h])
    # (2) permuting heads permutes outputs
    perm = torch.tensor([3, 0, 4, 1, 2], device="cuda")
    o_p, l_p = fwd(qd[:, perm].contiguous(), kd[:, perm].contiguous(), vd[:, perm].contiguous(),
                                          is_causal=True)
    assert torch.equal(o_p, o_all[:, perm]) and torch.equal(l_p, l_all[:, perm])
    # (3) padded batch/head strides (binding table slots 7,8: kernels.metal:608-609)
    def padded(x):
        buf = torch.zeros(B, H + 1, N + 8, D, dtype=x.dtype, device="cuda")
        view = buf[:, :H, :N, :]
        view.copy_(x)
        return view
    qp, kp, vp = padded(qd), padded(kd), padded(vd)
    assert qp.stride() == ((H + 1) * (N + 8) * D, (N + 8) * D, D, 1)
    o_s, l_s = fwd(qp, kp, vp, is_causal=True)
    assert torch.equal(o_s, o_all) and torch.equal(l_s, l_all)
    torch.cuda.synchronize()


@pytest.mark.parametrize("variant", MFMA_VARIANTS)
def test_randomized_shapes(fa, oracle_mod, variant):
    # seeded random (B, H, N, D, dtype, causal, scale): ragged N everywhere, both head dims, custom scales
    rng = np.random.default_rng(2024)
    for _ in range(60):
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 6))
        N = int(rng.choice([1, 2, 31, 33, 64, 96, 127, 128, 191, 257, 300, 449, 640]))
        D = int(rng.choice([32, 64, 96, 128, 256]))
        dtype = str(rng.choice(["f16", "bf16", "fp8"]))
        causal = bool(rng.integers(0, 2))
        scale = float(rng.choice([D ** -0.5, 0.05, 0.2]))
        seeds = tuple(int(x) for x in rng.integers(1, 10 ** 6, 3))
        if not fa.supported({"fp8": "fp8_e4m3"}.get(dtype, dtype), variant, D):
            continue
        q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype, seeds=seeds)
        o, lse = run_op(fa, q, k, v, dtype, causal, variant, scale)
        o64, l64 = oracle_mod.attn_fwd_f64(q, k, v, causal, scale)
        tol = (TOL_O["f16"] if dtype == "f16" else TOL_O["bf16"]) + fp8pv_term(variant, dtype, v)
        pre = is_prescaled(fa, dtype, variant, B, H, N, D, causal)
        assert np.abs(o - o64).max() < tol, (B, H, N, D, dtype, causal, scale)
        assert np.abs(lse - l64).max() < lse_tol(dtype, pre, q, k, scale) + fp8pv_lse_term(variant, dtype), (B, H, N, D, dtype, causal, scale)
        if pre:
            o64, l64 = oracle_mod.attn_fwd_f64(effective_q(oracle_mod, q, dtype, scale), k, v, causal, LN2)
            assert np.abs(o - o64).max() < tol and np.abs(lse - l64).max() < 1e-4 + rowsum_term(dtype, pre), (B, H, N, D, dtype, causal, scale)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_generalised_forward_gqa_and_rectangular(fa, oracle_mod, dtype):
    # scope row f3: grouped/multi-query heads and Nq != Nk (bottom-right causal alignment), vs the fp64 oracle
    import torch

    rng = np.random.default_rng(11)
    cases = [  # B, Hq, Hkv, Nq, Nk, D, causal
        (2, 8, 2, 200, 200, 64, True), (1, 8, 1, 130, 130, 64, False), (2, 4, 4, 64, 300, 64, True),
        (1, 6, 3, 1, 257, 64, True), (1, 4, 2, 100, 37, 64, False), (1, 8, 2, 129, 512, 128, True),
        (1, 2, 1, 77, 77, 128, True), (1, 16, 4, 33, 1000, 64, True), (1, 4, 2, 70, 150, 32, True),
        (1, 4, 1, 65, 65, 96, False), (1, 2, 2, 40, 300, 256, True),
        (2, 32, 8, 1, 1000, 128, True), (1, 8, 8, 1, 65, 64, False), (4, 32, 8, 16, 700, 64, True),  # decode steps / short chunks
        (3, 32, 4, 130, 260, 64, True), (1, 72, 8, 128, 128, 128, True)]  # more than 64 blocks: the 128-row kernel
    for (B, Hq, Hkv, Nq, Nk, D, causal) in cases:
        q = oracle_mod.round_to(oracle_mod.init_random(B * Hq * Nq * D, int(rng.integers(1, 1 << 20))).reshape(B, Hq, Nq, D), dtype)
        k = oracle_mod.round_to(oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), dtype)
        v = oracle_mod.round_to(oracle_mod.init_random(B * Hkv * Nk * D, int(rng.integers(1, 1 << 20))).reshape(B, Hkv, Nk, D), dtype)
        qd, kd, vd = (to_dev(x, dtype) for x in (q, k, v))
        o64, l64 = oracle_mod.attn_fwd_ex_f64(q, k, v, causal)
        o64q, l64q = oracle_mod.attn_fwd_ex_f64(effective_q(oracle_mod, q, dtype), k, v, causal, LN2)
        # which kernel fa_fwd_ex's AUTO runs (include/fa_mi355.h): the split-KV kernel for at most 64 blocks of 128 query rows against
        # more than 64 keys at head_dim 64 (it scales every score in fp32), else the 128-row kernel (pre-scaled operand) -- its
        # 16x16x32 form for long key sequences on large head_dim-64 grids; every kernel that takes the generalised problem also by name
        blocks = B * Hq * ((Nq + 127) // 128)
        small = D == 64 and Nk > 64 and blocks <= 64
        auto = "mfma_splitkv" if small else "mfma16" if (D in (64, 128) and Nk >= (2048 if D == 64 else 8192) and blocks > 512) else "mfma"
        for variant in ["auto", "mfma", "mfma_exact"] + (["mfma16"] if D in (64, 128) else []) + (["mfma_splitkv"] if D == 64 else []):
            name = auto if variant == "auto" else variant
            pre = {"mfma": int(D <= 128), "mfma16": 2}.get(name, 0)
            o, lse = fa.flash_attention_forward(qd, kd, vd, is_causal=causal, variant=variant)
            torch.cuda.synchronize()
            what = (variant, B, Hq, Hkv, Nq, Nk, D, causal)
            assert np.abs(o.float().cpu().numpy() - o64).max() < TOL_O[dtype], what
            assert np.abs(lse.cpu().numpy() - l64).max() < lse_tol(dtype, pre, q, k), what
            if pre:  # strict vs the oracle on the operand the kernel really multiplies
                assert np.abs(o.float().cpu().numpy() - o64q).max() < TOL_O[dtype], what
                assert np.abs(lse.cpu().numpy() - l64q).max() < 1e-4 + rowsum_term(dtype, pre), what
            if Nq == Nk:  # GQA == MHA on repeated K/V heads, bit for bit, through the same kernel by name
                g = Hq // Hkv
                o2, l2 = fa.flash_attention_forward(qd, kd.repeat_interleave(g, 1).contiguous(), vd.repeat_interleave(g, 1).contiguous(),
                                                    is_causal=causal, variant=name)
                assert torch.equal(o, o2) and torch.equal(lse, l2), what
    # a kernel that does not take grouped heads / Nq != Nk is refused by name, never silently replaced
    x = to_dev(np.zeros((1, 4, 64, 64), np.float32), dtype)
    for variant in ("mfma_pp", "mfma_split2", "mfma_h64s2", "tiled_v2", "naive"):
        with pytest.raises(fa.FaError) as e:
            fa.flash_attention_forward(x, x[:, :2].contiguous(), x[:, :2].contiguous(), variant=variant)
        assert e.value.status == -2
    if dtype == "bf16":  # the same generalised path with fp8 (e4m3) inputs: equal to bf16 on the same values up to fp32 summation order
        q = oracle_mod.round_to(oracle_mod.init_random(2 * 8 * 96 * 64, 5).reshape(2, 8, 96, 64) * 2, "fp8")
        k = oracle_mod.round_to(oracle_mod.init_random(2 * 2 * 333 * 64, 6).reshape(2, 2, 333, 64) * 2, "fp8")
        v = oracle_mod.round_to(oracle_mod.init_random(2 * 2 * 333 * 64, 7).reshape(2, 2, 333, 64) * 2, "fp8")
        o8, l8 = fa.flash_attention_forward(to_dev(q, "fp8"), to_dev(k, "fp8"), to_dev(v, "fp8"), is_causal=True)
        ob, lb = fa.flash_attention_forward(to_dev(q, "bf16"), to_dev(k, "bf16"), to_dev(v, "bf16"), is_causal=True)
        assert (l8 - lb).abs().max().item() < lse_tol("bf16", True, q, k, None, 5e-5)
        assert (o8.float() - ob.float()).abs().max().item() <= 2 ** -6 * ob.float().abs().max().item()
        o64, _ = oracle_mod.attn_fwd_ex_f64(q, k, v, True)
        assert np.abs(o8.float().cpu().numpy() - o64).max() < TOL_O["bf16"] * 2
    x = to_dev(np.zeros((1, 4, 64, 64), np.float32), dtype)
    with pytest.raises(fa.FaError):  # causal with fewer keys than queries would leave empty rows: refused
        fa.flash_attention_forward(x, x[:, :2, :32].contiguous(), x[:, :2, :32].contiguous(), is_causal=True)
    with pytest.raises(ValueError):  # Hq not a multiple of Hkv
        fa.flash_attention_forward(x, x[:, :3].contiguous(), x[:, :3].contiguous())


def test_reentrant_across_streams(fa, oracle_mod):
    # include/fa_mi355.h: "no state, re-entrant across devices/streams": two different problems launched
    # concurrently on two streams give the same bits as when run alone
    import torch

    a = [to_dev(x, "bf16") for x in make_qkv(oracle_mod, 2, 8, 1024, 64, "bf16", seeds=(1, 2, 3))]
    b = [to_dev(x, "f16") for x in make_qkv(oracle_mod, 1, 4, 777, 128, "f16", seeds=(4, 5, 6))]
    ra = fa.flash_attention_forward(*a, is_causal=True)
    rb = fa.flash_attention_forward(*b, is_causal=False)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(5):
        with torch.cuda.stream(s1):
            oa = fa.flash_attention_forward(*a, is_causal=True)
        with torch.cuda.stream(s2):
            ob = fa.flash_attention_forward(*b, is_causal=False)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa[0], ra[0]) and torch.equal(oa[1], ra[1])
        assert torch.equal(ob[0], rb[0]) and torch.equal(ob[1], rb[1])


def test_launch_is_graph_capturable(fa, oracle_mod):
    # the C-ABI promises no allocation / synchronisation inside the launch (HIPGUIDE guideline 9):
    # capture forward + backward into a HIP graph, replay it on new data, compare with eager launches
    import torch

    B, H, N, D = 2, 4, 512, 64
    q, k, v = (to_dev(x, "bf16") for x in make_qkv(oracle_mod, B, H, N, D, "bf16"))
    do = to_dev(oracle_mod.round_to(oracle_mod.init_random(B * H * N * D, 45).reshape(B, H, N, D), "bf16"), "bf16")
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
    lib = fa.load_library()
    dq, dk, dv = (torch.empty(B, H, N, D, dtype=torch.float32, device="cuda") for _ in range(3))
    ws = torch.empty(lib.fa_bwd_workspace_bytes(B, H, N), dtype=torch.uint8, device="cuda")

    def launch(stream):
        assert lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, D, D ** -0.5,
                          H * N * D, N * D, 1, 2, 0, stream) == 0
        assert lib.fa_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                          dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, H, N, D, D ** -0.5, H * N * D, N * D,
                          1, 2, stream) == 0

    launch(torch.cuda.current_stream().cuda_stream)  # eager reference (also warms the module)
    torch.cuda.synchronize()
    ref = [t.clone() for t in (o, lse, dq, dk, dv)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch(torch.cuda.current_stream().cuda_stream)
    for t in (o, lse, dq, dk, dv):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip((o, lse, dq, dk, dv), ref):
        assert torch.equal(a, b)
    q.copy_(q.flip(2))  # new inputs, same graph
    g.replay()
    torch.cuda.synchronize()
    launch(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not torch.equal(o, ref[0])


def test_forward_plan_matches_direct_call(fa, oracle_mod):
    # ForwardPlan = the checks of flash_attention_forward once + bare launches: same bits, sees new data in the same
    # tensors, and rejects what the direct call rejects
    import torch

    for dtype, causal, shape in (("bf16", True, (2, 3, 200, 64)), ("f16", False, (1, 8, 1024, 64)), ("bf16", True, (1, 2, 300, 128))):
        q, k, v = (to_dev(x, dtype) for x in make_qkv(oracle_mod, *shape, dtype))
        o_ref, lse_ref = fa.flash_attention_forward(q, k, v, is_causal=causal)
        plan = fa.ForwardPlan(q, k, v, is_causal=causal)
        for _ in range(3):
            plan.launch()
        torch.cuda.synchronize()
        assert torch.equal(plan.out, o_ref) and torch.equal(plan.lse, lse_ref)
        q.copy_(q.flip(2))
        plan.launch()
        o2, lse2 = fa.flash_attention_forward(q, k, v, is_causal=causal)
        torch.cuda.synchronize()
        assert torch.equal(plan.out, o2) and torch.equal(plan.lse, lse2) and not torch.equal(o2, o_ref)
    with pytest.raises(ValueError):
        fa.ForwardPlan(q, k, v, out=torch.empty(1, 2, 300, 64, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(ValueError):
        fa.ForwardPlan(q, k[:, :, :10], v[:, :, :10], is_causal=True)  # Nq != Nk goes through flash_attention_forward


def test_torch_custom_op_matches_sdpa(fa, oracle_mod):
    # the kernel as a torch operator, compared in-process with torch's own attention
    import torch
    import torch.nn.functional as F

    from flash_attention_metal_amd import torch_op  # registers torch.ops.fa_mi355.attention_forward

    q, k, v = (to_dev(x, "bf16") for x in make_qkv(oracle_mod, 2, 4, 384, 64, "bf16"))
    for causal in (False, True):
        o, lse = torch.ops.fa_mi355.attention_forward(q, k, v, causal, 0.0)
        ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float(), is_causal=causal)
        assert (o.float() - ref).abs().max().item() < TOL_O["bf16"]
        o2, lse2 = fa.flash_attention_forward(q, k, v, is_causal=causal)
        assert torch.equal(o, o2) and torch.equal(lse, lse2)
    meta = torch.ops.fa_mi355.attention_forward(q.to("meta"), k.to("meta"), v.to("meta"), True, 0.0)
    assert meta[0].shape == q.shape and meta[1].shape == (2, 4, 384) and meta[1].dtype == torch.float32
    with pytest.raises(Exception):  # no CPU kernel is registered
        torch.ops.fa_mi355.attention_forward(q.cpu(), k.cpu(), v.cpu(), False, 0.0)
    torch.cuda.synchronize()


def test_error_behaviour_on_device(fa):
    import torch

    x = torch.zeros(1, 1, 128, 136, dtype=torch.bfloat16, device="cuda")  # (every multiple of 8 up to 128 has a kernel; between 128 and 256 none)
    with pytest.raises(fa.FaError) as e:
        fa.flash_attention_forward(x, x, x)
    assert e.value.status == -2
    y = torch.zeros(1, 1, 128, 64, dtype=torch.float32, device="cuda")
    with pytest.raises(fa.FaError):
        fa.flash_attention_forward(y, y, y, variant="mfma")
    with pytest.raises(fa.FaError):
        fa.flash_attention_forward(y.bfloat16(), y.bfloat16(), y.bfloat16(), scale=-1.0)
    with pytest.raises(ValueError):
        fa.flash_attention_forward(y, y[:, :, :64], y)
    # caller-supplied out / lse are validated on BOTH entry points (square and generalised): wrong strides, dtype,
    # size or device would make the kernel write out of place
    q = torch.zeros(2, 4, 128, 64, dtype=torch.bfloat16, device="cuda")
    kv = torch.zeros(2, 2, 256, 64, dtype=torch.bfloat16, device="cuda")
    qs = torch.zeros(2, 4, 136, 64, dtype=torch.bfloat16, device="cuda")[:, :, :128]  # strided q
    for qq, kk in ((q, q), (q, kv), (qs, kv)):
        with pytest.raises(ValueError):
            fa.flash_attention_forward(qq, kk, kk, out=torch.empty(2, 4, 128, 64, dtype=torch.float16, device="cuda"))
        with pytest.raises(ValueError):
            fa.flash_attention_forward(qq, kk, kk, out=torch.empty(2, 4, 64, 64, dtype=torch.bfloat16, device="cuda"))
        with pytest.raises(ValueError):
            fa.flash_attention_forward(qq, kk, kk, lse=torch.empty(2, 4, 64, dtype=torch.float32, device="cuda"))
        with pytest.raises(ValueError):
            fa.flash_attention_forward(qq, kk, kk, lse=torch.empty(2, 4, 128, dtype=torch.float16, device="cuda"))
        with pytest.raises(ValueError):
            fa.flash_attention_forward(qq, kk, kk, out=torch.empty(2, 4, 128, 64, dtype=torch.bfloat16))  # CPU tensor
    with pytest.raises(ValueError):  # contiguous out for a strided q
        fa.flash_attention_forward(qs, kv, kv, out=torch.empty(2, 4, 128, 64, dtype=torch.bfloat16, device="cuda"))
    torch.cuda.synchronize()


# --------------------------------------------------------------------------
# BASELINE.json configurations at FULL size: sampled rows vs the fp64 oracle +
# size-independent properties
# --------------------------------------------------------------------------
def test_auto_routes_reach_every_kernel_and_match_the_oracle(fa, oracle_mod):
    # FA_VARIANT_AUTO picks among the matrix-core kernels by the grid a shape gives (csrc/fa_api.hip fa_resolve_variant_for;
    # DESIGN.md section 4). One shape on each side of every boundary, called THROUGH auto: the resolved variant is what
    # DESIGN says, the result equals that of the kernel called by name bit for bit, and it matches the oracle.
    import torch

    lib = fa.load_library()
    V = fa.VARIANTS
    cases = [  # (B, H, N, D, dtype, causal, expected variant)
        (1, 8, 1024, 64, "f16", False, "mfma_splitkv"),    # BASELINE config 2: 64 blocks of 128 rows <= 64
        (1, 8, 1040, 64, "bf16", True, "mfma_h64s2"),      # 72 blocks of 128 rows: just past the split-KV rule; 136 blocks of 64 rows
        (1, 64, 512, 64, "bf16", True, "mfma_h64s2"),      # 512 blocks of 64 rows: the last grid of the 64-row two-split form
        (1, 65, 512, 64, "bf16", False, "mfma"),           # 520 blocks of 64 rows, 260 of 128: first grid of the plain 128-row kernel
        (1, 80, 256, 64, "bf16", True, "mfma_h64s2"),      # 320 blocks of 64 rows, N >= 256
        (1, 200, 128, 64, "f16", True, "mfma"),            # N < 256: plain kernel
        (1, 40, 512, 64, "fp8", True, "mfma_split2"),      # fp8 keeps the eight-wave form (160 blocks of 128 rows)
        (2, 3, 64, 64, "f16", True, "mfma"),               # N <= 64: a single tile, never split-KV
        (1, 2, 4096, 128, "bf16", True, "mfma_split2"),    # head_dim 128, 64 blocks: never split-KV (its head_dim-128 build spills)
        (1, 8, 256, 128, "bf16", False, "mfma"),           # head_dim 128, 16 blocks, N < 512: the plain kernel
        (1, 16, 2048, 128, "bf16", True, "mfma_split2"),   # 256 blocks of 128 rows, N < 4096
        (1, 32, 2048, 64, "bf16", True, "mfma_split2"),    # causal, 512 blocks, N >= 2048: still the eight-wave form
        (1, 32, 2048, 64, "bf16", False, "mfma16"),        # ... not without the mask: head_dim 64, 16-bit inputs, N >= 2048 -> the 16x16x32 kernel
        (1, 80, 2048, 64, "f16", True, "mfma16"),          # causal, 1280 blocks of 128 rows (past the eight-wave form), N >= 2048
        (1, 80, 1535, 64, "bf16", True, "mfma"),           # ... N < 1536: the 32x32x16 kernel
        (1, 80, 1536, 64, "bf16", True, "mfma16"),
        (1, 128, 1024, 64, "bf16", False, "mfma16"),       # without the mask from N = 512 on, for grids of at least 512 workgroups of 128 rows
        (1, 256, 512, 64, "f16", False, "mfma16"),
        (1, 255, 511, 64, "bf16", False, "mfma"),
        (1, 80, 2048, 64, "fp8", True, "mfma_fp8pv"),      # ... fp8 inputs have no 16x16x32 kernel: the all-fp8 kernel on grids that fill the chip
        (1, 64, 1024, 64, "bf16", True, "mfma"),           # ... and not at N = 1024
        (1, 64, 1024, 128, "bf16", True, "mfma"),          # head_dim-128 causal sequences below 2048: the 32x32x16 128-row kernel
        (1, 64, 2048, 128, "bf16", True, "mfma16"),        # ... from 2048 on (without the mask from 1024 on) its 16x16x32 form
        (1, 64, 1024, 128, "f16", False, "mfma16"),
        (1, 32, 2048, 128, "fp8", True, "mfma_fp8pv"),     # e4m3 at head_dim 128: the all-fp8 kernel also on grids the eight-wave form takes at 64
        (1, 8, 4096, 128, "fp8", True, "mfma_split2"),     # ... except small grids of long sequences
        (1, 16, 8192, 128, "bf16", True, "mfma16"),        # (config 4's shape family)
        (1, 4, 300, 96, "bf16", True, "mfma"),             # head dims only the 128-row kernel has
        (1, 8, 1024, 64, "fp8", True, "mfma_splitkv"),
        (1, 40, 1024, 64, "fp8", True, "mfma_split2"),     # fp8, causal, 320 blocks, N >= 1024: the eight-wave form
        (1, 80, 1024, 64, "fp8", True, "mfma_fp8pv"),      # 640 blocks: past the eight-wave form, the all-fp8 kernel
        (1, 80, 1024, 128, "fp8", True, "mfma_fp8pv"),     # ... at head_dim 128 too
        (1, 8, 512, 256, "fp8", True, "mfma"),             # head_dim 256: the 128-row kernel with bf16 probabilities
    ]
    for (B, H, N, D, dtype, causal, want) in cases:
        fdt = fa.DTYPES[{"fp8": "fp8_e4m3"}.get(dtype, dtype)]
        assert lib.fa_resolve_variant_for(fdt, D, B, H, N, int(causal)) == V[want], (B, H, N, D, dtype, causal, want)
        q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype, seeds=(5, 6, 7))
        qd, kd, vd = (to_dev(x, dtype) for x in (q, k, v))
        o_a, l_a = fa.flash_attention_forward(qd, kd, vd, is_causal=causal)  # variant="auto"
        o_n, l_n = fa.flash_attention_forward(qd, kd, vd, is_causal=causal, variant=want)
        torch.cuda.synchronize()
        assert torch.equal(o_a, o_n) and torch.equal(l_a, l_n), (want, "auto did not run the kernel it resolves to")
        hs = [0, H - 1] if N > 1024 else list(range(H))  # the oracle is O(N^2) per head
        for h in hs:
            rows = np.unique(np.concatenate([[0, N // 2, N - 1], np.arange(0, N, max(1, N // 64))])).astype(np.int32)
            o64, l64 = oracle_mod.attn_rows_f64(q[0, h], k[0, h], v[0, h], rows, causal)
            pre = is_prescaled(fa, dtype, "auto", B, H, N, D, causal)
            tol = (TOL_O["f16"] if dtype == "f16" else TOL_O["bf16"]) + fp8pv_term(want, dtype, v[0, h])
            assert np.abs(o_a[0, h].float().cpu().numpy()[rows] - o64).max() < tol, (want, h)
            assert np.abs(l_a[0, h].cpu().numpy()[rows] - l64).max() < lse_tol(dtype, pre, q[0, h], k[0, h]) + fp8pv_lse_term(want, dtype), (want, h)



def _full_size(fa, oracle_mod, B, H, N, D, dtype, causal, heads, nrows=48, variant="auto"):
    import functools

    import torch

    fwd = functools.partial(fa.flash_attention_forward, variant=variant)

    g = torch.Generator(device="cuda").manual_seed(1234)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    q, k, v = (torch.rand(B, H, N, D, generator=g, device="cuda", dtype=torch.float32).mul_(2).sub_(1).to(tdt)
               for _ in range(3))
    o, lse = fwd(q, k, v, is_causal=causal)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    pre = is_prescaled(fa, dtype, variant, B, H, N, D, causal)
    if causal:  # row 0 attends to key 0 only
        assert torch.equal(o[:, :, 0], v[:, :, 0])
        s00 = (q[:, :, 0].float() * k[:, :, 0].float()).sum(-1) * (D ** -0.5)
        assert torch.allclose(lse[:, :, 0], s00, atol=lse_tol(dtype, pre, q[:, :, :1].float().cpu().numpy(), k[:, :, :1].float().cpu().numpy()))
    rng = np.random.default_rng(7)
    worst_o = worst_l = tol_l = 0.0
    for (b, h) in heads:
        rows = np.unique(np.concatenate([[0, 1, 31, 32, 63, 64, 127, 128, N - 129, N - 128, N - 65, N - 64, N - 1],
                                         rng.integers(0, N, nrows)])).astype(np.int32)
        qh, kh, vh = (x[b, h].float().cpu().numpy() for x in (q, k, v))
        o64, l64 = oracle_mod.attn_rows_f64(qh, kh, vh, rows, causal)
        worst_o = max(worst_o, np.abs(o[b, h].float().cpu().numpy()[rows] - o64).max())
        worst_l = max(worst_l, np.abs(lse[b, h].cpu().numpy()[rows] - l64).max())
        tol_l = max(tol_l, lse_tol(dtype, pre, qh, kh, None, TOL_LSE[dtype]))
        if pre:  # strict vs the oracle on the operand the kernel multiplies
            o64, l64 = oracle_mod.attn_rows_f64(effective_q(oracle_mod, qh, dtype), kh, vh, rows, causal, LN2)
            assert np.abs(o[b, h].float().cpu().numpy()[rows] - o64).max() < TOL_O[dtype]
            assert np.abs(lse[b, h].cpu().numpy()[rows] - l64).max() < TOL_LSE[dtype] + rowsum_term(dtype, pre)
    assert worst_o < TOL_O[dtype] and worst_l < tol_l, (worst_o, worst_l)
    # V = const -> O = const (the softmax weights sum to 1), any size
    ones = torch.full_like(v, 0.5)
    o1, _ = fwd(q, k, ones, is_causal=causal)
    assert (o1.float() - 0.5).abs().max().item() <= 0.5 * 2 ** -7
    # rerun is deterministic bit for bit
    o2, lse2 = fwd(q, k, v, is_causal=causal)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    torch.cuda.synchronize()


@pytest.mark.parametrize("D", [8, 16, 24, 40, 48, 56, 72, 80, 88, 104, 112, 120])
def test_forward_any_multiple_of_eight(fa, oracle_mod, D):
    """Head dims without a kernel of their own (scope row f3): every multiple of 8 up to 128 runs the 16x16x32 kernel's next larger
    instantiation on zero-padded rows (csrc/fa_mfma16_kernel.hip, PAD) -- through AUTO, both 16-bit dtypes, causal and full, ragged
    lengths with partial blocks and tiles, and through fa_fwd_ex with grouped heads and Nq != Nk."""
    import torch

    assert fa.supported("bf16", "auto", D) and fa.load_library().fa_resolve_variant_for(2, D, 1, 2, 300, 1) == fa.VARIANTS["mfma16"]
    for dtype in ("bf16", "f16"):
        for (B, H, N) in ((1, 2, 300), (2, 3, 129), (1, 1, 1), (1, 2, 1000)):
            q, k, v = make_qkv(oracle_mod, B, H, N, D, dtype)
            for causal in (False, True):
                check(fa, oracle_mod, q, k, v, dtype, causal, "auto", tol_scale=2.0)
    # grouped heads and a longer key sequence (bottom-right aligned mask)
    dtype = "bf16"
    q, _, _ = make_qkv(oracle_mod, 1, 4, 100, D, dtype)
    _, k, v = make_qkv(oracle_mod, 1, 2, 260, D, dtype)
    for causal in (False, True):
        o, lse = fa.flash_attention_forward(to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype), is_causal=causal)
        torch.cuda.synchronize()
        ke, ve = (np.repeat(x, 2, axis=1) for x in (k, v))
        s_ = np.einsum("bhid,bhjd->bhij", q.astype(np.float64), ke.astype(np.float64)) / np.sqrt(D)
        if causal:
            i, j = np.arange(100)[:, None], np.arange(260)[None, :]
            s_ = np.where(j <= i + 160, s_, -np.inf)
        m_ = s_.max(-1, keepdims=True)
        p_ = np.exp(s_ - m_)
        o64 = np.einsum("bhij,bhjd->bhid", p_ / p_.sum(-1, keepdims=True), ve.astype(np.float64))
        l64 = (m_ + np.log(p_.sum(-1, keepdims=True)))[..., 0]
        assert o.shape == (1, 4, 100, D)
        assert np.abs(o.float().cpu().numpy() - o64).max() < TOL_O[dtype] * 2
        assert np.abs(lse.cpu().numpy() - l64).max() < lse_tol(dtype, 2, q, np.repeat(k, 2, axis=1))


def test_mfma16_eight_wave_workgroups(fa, oracle_mod):
    """The 16x16x32 kernel runs 256-row workgroups (eight waves sharing every K / V tile) where its launch rule says so
    (csrc/fa_mfma16_kernel.hip, mfma16_waves): non-causal grids of at least 512 such workgroups, the two-round causal grids of config 3's
    kind, long non-causal head_dim-128 grids. Ragged lengths on both sides of the 256-row blocks, both dtypes, rows sampled across block
    and wave boundaries. (The whole mfma16 parity suite was also run once with eight waves forced on EVERY shape: FA16_FORCE_RW=8 build,
    28 tests, profiles/r04/tests_mfma16_eight_waves_forced.log.)"""
    lib = fa.load_library()
    # which instantiation AUTO launches (the kernel name carries the waves per workgroup)
    assert lib.fa_fwd_kernel_name(2, 64, 4, 16, 4096, 1).decode().endswith("64, true, 8, false, 1>")    # config 3
    assert lib.fa_fwd_kernel_name(2, 64, 4, 16, 4100, 1).decode().endswith("64, true, 8, false, 1>")
    assert lib.fa_fwd_kernel_name(2, 64, 8, 16, 4096, 1).decode().endswith("64, true, 4, false, 1>")    # twice the heads: four waves
    assert lib.fa_fwd_kernel_name(2, 64, 4, 16, 2048, 1).decode().endswith("64, true, 4, false, 1>")
    assert lib.fa_fwd_kernel_name(2, 64, 4, 16, 2048, 0).decode().endswith("64, false, 8, false, 1>")
    assert lib.fa_fwd_kernel_name(2, 128, 1, 32, 8192, 0).decode().endswith("128, false, 8, false, 1>")
    assert lib.fa_fwd_kernel_name(2, 128, 1, 32, 16384, 1).decode().endswith("128, true, 4, false, 1>")  # config 4's shard
    _full_size(fa, oracle_mod, 1, 128, 1000, 64, "bf16", False, heads=[(0, 0), (0, 127)], variant="mfma16")   # 512 workgroups of 256 rows, ragged
    _full_size(fa, oracle_mod, 4, 16, 4100, 64, "f16", True, heads=[(0, 0), (3, 15)], variant="mfma16")       # 1088 workgroups, causal, ragged
    _full_size(fa, oracle_mod, 4, 16, 2304, 64, "bf16", False, heads=[(1, 7)], variant="auto")                # AUTO's non-causal route
    _full_size(fa, oracle_mod, 1, 32, 8192, 128, "bf16", False, heads=[(0, 5)], nrows=24, variant="mfma16")   # head_dim 128
    # grouped heads and Nq != Nk on the eight-wave form (fa_fwd_ex: 256 query heads on 32 key heads, 500 queries x 2000 keys, no mask)
    import torch

    g = torch.Generator(device="cuda").manual_seed(77)
    q = torch.rand(1, 256, 500, 64, generator=g, device="cuda").mul_(2).sub_(1).to(torch.bfloat16)
    k, v = (torch.rand(1, 32, 2000, 64, generator=g, device="cuda").mul_(2).sub_(1).to(torch.bfloat16) for _ in range(2))
    o, lse = fa.flash_attention_forward(q, k, v, is_causal=False, variant="mfma16")
    torch.cuda.synchronize()
    rows = np.array([0, 1, 31, 32, 127, 128, 255, 256, 257, 383, 384, 499], dtype=np.int32)
    for h in (0, 7, 8, 255):
        qh, kh, vh = q[0, h].float().cpu().numpy(), k[0, h // 8].float().cpu().numpy(), v[0, h // 8].float().cpu().numpy()
        s_ = (qh[rows].astype(np.float64) @ kh.astype(np.float64).T) / 8.0
        m_ = s_.max(-1, keepdims=True)
        p_ = np.exp(s_ - m_)
        o64 = (p_ / p_.sum(-1, keepdims=True)) @ vh.astype(np.float64)
        l64 = (m_ + np.log(p_.sum(-1, keepdims=True)))[:, 0]
        assert np.abs(o[0, h].float().cpu().numpy()[rows] - o64).max() < TOL_O["bf16"], h
        assert np.abs(lse[0, h].cpu().numpy()[rows] - l64).max() < lse_tol("bf16", 2, qh, kh), h


@pytest.mark.parametrize("variant", ["auto", "tiled_v2", "mfma", "mfma_splitkv", "mfma_split2", "mfma_exact", "mfma_h64s2", "mfma16"])
def test_config2_full(fa, oracle_mod, variant):  # seqlen=1024, D=64, B=1, H=8, fp16, non-causal
    # BASELINE configs[1] names the "V2-style tiled kernel": variant tiled_v2 (kernels.metal:462-596) runs it at
    # full size in fp16; the matrix-core kernels are checked on the same tensors
    _full_size(fa, oracle_mod, 1, 8, 1024, 64, "f16", False, [(0, 0), (0, 7)], variant=variant)


@pytest.mark.parametrize("variant", ["auto", "mfma", "mfma_exact", "mfma16"])
def test_config3_full(fa, oracle_mod, variant):  # seqlen=4096, D=64, B=4, H=16, bf16, causal
    _full_size(fa, oracle_mod, 4, 16, 4096, 64, "bf16", True, [(0, 0), (1, 5), (3, 15)], variant=variant)


@pytest.mark.parametrize("variant", ["auto", "mfma_exact", "mfma_fp8pv"])
def test_config5_full_fp8(fa, oracle_mod, variant):  # seqlen=8192, D=64, fp8 in / fp32 acc, causal (B=4,H=16 assumed)
    import torch

    B, H, N, D = 4, 16, 8192, 64
    g = torch.Generator(device="cuda").manual_seed(99)
    q, k, v = (torch.rand(B, H, N, D, generator=g, device="cuda").mul_(2).sub_(1).to(torch.float8_e4m3fn) for _ in range(3))
    o, lse = fa.flash_attention_forward(q, k, v, is_causal=True, variant=variant)
    torch.cuda.synchronize()
    assert o.dtype == torch.bfloat16 and torch.isfinite(o).all() and torch.isfinite(lse).all()
    assert torch.equal(o[:, :, 0], v[:, :, 0].to(torch.bfloat16))  # causal row 0 == V[0]
    # which kernel ran: the all-fp8 one (by name, or where AUTO resolves to it) rounds the probabilities to e4m3
    name = variant if variant != "auto" else {v_: k_ for k_, v_ in fa.VARIANTS.items()}[fa.load_library().fa_resolve_variant_for(3, D, B, H, N, 1)]
    rng = np.random.default_rng(3)
    errs = []
    for (b, h) in ((0, 0), (3, 15)):
        rows = np.unique(np.concatenate([[0, 1, 63, 64, 127, 128, N - 65, N - 64, N - 1], rng.integers(0, N, 32)])).astype(np.int32)
        qh, kh, vh = (x[b, h].float().cpu().numpy() for x in (q, k, v))
        o64, l64 = oracle_mod.attn_rows_f64(qh, kh, vh, rows, True)
        err = np.abs(o[b, h].float().cpu().numpy()[rows] - o64)
        assert err.max() < TOL_O["bf16"] + fp8pv_term(name, "fp8", vh), (name, err.max())
        lerr = np.abs(lse[b, h].cpu().numpy()[rows] - l64)
        assert lerr.max() < TOL_LSE["bf16"] + fp8pv_lse_term(name, "fp8")
        assert lerr[rows > 64].max() < TOL_LSE["bf16"] + 0.1 * fp8pv_lse_term(name, "fp8"), (name, lerr[rows > 64].max())  # long rows: the roundings average out (measured 4.1e-3)
        errs.append(err[rows > 1024])
    # long rows (more than 1024 comparable keys): the independent roundings of the e4m3 probabilities average out -- the bf16 bar holds
    assert np.concatenate(errs).max() < TOL_O["bf16"], (name, np.concatenate(errs).max())


@pytest.mark.parametrize("variant", ["mfma16", "mfma"])
def test_config4_sharded_equals_unsharded_bit_for_bit(fa, oracle_mod, variant):
    # BASELINE configs[3] shards (batch, head) over 8 GPUs. The shards must be the SAME computation: here the
    # 8 shards of a reduced-batch config-4 tensor (B=8 -> 8 shards of 2 heads each, N=4096, D=128, bf16 causal) are
    # computed one by one on this GPU (as 8 ranks would, flash_attention_metal_amd.shard.shard_heads) and compared bit
    # for bit with the unsharded call. The kernels are named explicitly ("auto" on this reduced problem would pick by grid
    # size); config 4 itself runs "mfma16" on every rank.
    import torch

    from flash_attention_metal_amd.shard import shard_heads

    B, H, N, D, world = 8, 2, 4096, 128, 8
    g = torch.Generator(device="cuda").manual_seed(4)
    q, k, v = (torch.rand(B, H, N, D, generator=g, device="cuda").mul_(2).sub_(1).to(torch.bfloat16) for _ in range(3))
    o_all, l_all = fa.flash_attention_forward(q, k, v, is_causal=True, variant=variant)
    qf, kf, vf = (x.view(1, B * H, N, D) for x in (q, k, v))
    for rank in range(world):
        lo, hi = shard_heads(B * H, world, rank)
        o_r, l_r = fa.flash_attention_forward(qf[:, lo:hi].contiguous(), kf[:, lo:hi].contiguous(), vf[:, lo:hi].contiguous(),
                                              is_causal=True, variant=variant)
        assert torch.equal(o_r[0], o_all.view(B * H, N, D)[lo:hi]) and torch.equal(l_r[0], l_all.view(B * H, N)[lo:hi])
    lib = fa.load_library()  # and the real config 4 resolves to the same kernel sharded or not
    assert lib.fa_resolve_variant_for(2, 128, 8, 32, 16384, 1) == lib.fa_resolve_variant_for(2, 128, 1, 32, 16384, 1) == fa.VARIANTS["mfma16"]
    torch.cuda.synchronize()


def test_config4_per_gpu_slice_full(fa, oracle_mod):  # seqlen=16384, D=128, bf16 causal; one GPU's 32 (b,h) slices
    _full_size(fa, oracle_mod, 1, 32, 16384, 128, "bf16", True, [(0, 0), (0, 31)], nrows=24)
