"""Shared helpers for the parity tests (test side only: imports the oracle)."""
import numpy as np

TORCH_DTYPE = {"f32": "float32", "f16": "float16", "bf16": "bfloat16", "fp8": "float8_e4m3fn"}


def make_qkv(oracle, B, H, N, D, dtype, seeds=(42, 43, 44), amp=1.0):
    """Independent Q,K,V ~ U(-1,1) (main.mm:24-30 generator, seeds 42/43/44), rounded RNE to dtype.

    Returns fp32 numpy arrays holding exactly-representable values of `dtype`."""
    out = []
    for s in seeds:
        x = oracle.init_random(B * H * N * D, s).reshape(B, H, N, D) * np.float32(amp)
        out.append(oracle.round_to(x, dtype))
    return out


def to_dev(x, dtype):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x)).to(getattr(torch, TORCH_DTYPE[dtype])).cuda()


def run_op(fa, q, k, v, dtype, causal, variant="auto", scale=None):
    import torch

    o, lse = fa.flash_attention_forward(to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype),
                                        is_causal=causal, variant=variant, scale=scale)
    torch.cuda.synchronize()
    return o.float().cpu().numpy(), lse.cpu().numpy()



# ---- the pre-scaled query operand (csrc/fa_mfma_kernel.hip; include/fa_mi355.h "LSE accuracy") ---------------------
# The 128-row matrix-core kernel and its eight-wave form multiply Q by scale*log2(e) in fp32 and round the product to the
# input type once per block: they compute the EXACT operator on that Q~ (with scale ln 2). The tests therefore hold them
# to the strict tolerances against the oracle evaluated on Q~ (effective_q below reproduces the kernel's two fp32
# multiplications and its RNE rounding bit for bit), and to the documented bound against the oracle on the true Q.
PRESCALE_EPS = {"f16": 2.0 ** -12, "bf16": 2.0 ** -9}  # half an ulp of the input type, relative
LN2 = 0.6931471805599453


def is_prescaled(fa, dtype, variant, B, H, N, D, causal=False):
    """Does fa_fwd(variant) run a kernel with the pre-scaled operand for this problem?"""
    if dtype not in PRESCALE_EPS or D > 128:
        return False
    v = fa.VARIANTS[variant]
    if v == 0:
        v = fa.load_library().fa_resolve_variant_for(fa.DTYPES[dtype], D, B, H, N, int(causal))
    if v == fa.VARIANTS["mfma16"]:
        return 2  # pre-scaled operand AND row sums over the ROUNDED probabilities (rowsum_term below)
    return int(v in (fa.VARIANTS["mfma"], fa.VARIANTS["mfma_split2"], fa.VARIANTS["mfma_h64s2"]))


# The 16x16x32 kernel (csrc/fa_mfma16_kernel.hip) takes its row sums from the matrix core: l = sum of the probabilities AFTER their
# rounding to the input type (the same values the PV product multiplies, so O's weights add up to exactly 1), where the other kernels
# add the fp32 probabilities. ln(l) therefore carries the rounding of P: at most half an ulp of the input type, relative (every P moves
# by a factor within 1 +- eps), far less on average (the roundings of a row's many P values are independent). include/fa_mi355.h, "LSE accuracy".
ROWSUM_EPS = {"f16": 2.0 ** -11, "bf16": 2.0 ** -8}


# The all-fp8 kernel (csrc/fa_fp8_kernel.hip, variant mfma_fp8pv) rounds every probability to e4m3 (3 mantissa bits) for the PV product:
# each weight moves by a factor within 1 +- 2^-4, so |O - exact| <= 2^-4 * max|V| on top of the other kernels' bar (far less where a
# row has many comparable keys: the roundings are independent).
def fp8pv_term(variant, dtype, v):
    return 2.0 ** -4 * float(np.abs(v).max()) if (variant == "mfma_fp8pv" and dtype == "fp8") else 0.0


# ... and its row sum comes out of the matrix core too (ones x the e4m3 probabilities): l adds the ROUNDED weights, each within a factor
# 1 +- 2^-4 of the exact one, so |LSE - exact| <= ln(1 + 2^-4) < 2^-4 -- reached by rows with two or three comparable keys only; the
# independent roundings of a long row average out (measured on config 5: at most 4.1e-3 over rows with more than 64 keys).
def fp8pv_lse_term(variant, dtype):
    return 2.0 ** -4 if (variant == "mfma_fp8pv" and dtype == "fp8") else 0.0


def rowsum_term(dtype, prescaled):
    return ROWSUM_EPS.get(dtype, 0.0) if prescaled == 2 else 0.0


def effective_q(oracle, q, dtype, scale=None):
    """Q~ as the kernel forms it: fp32(scale) * fp32(log2 e) -> c; fp32(q) * c; RNE to dtype. Use with scale = LN2."""
    sc = np.float32(q.shape[-1] ** -0.5 if scale is None else scale)
    c2 = np.float32(sc * np.float32(1.4426950408889634))
    return oracle.round_to((q.astype(np.float32) * c2).astype(np.float32), dtype)


def prescale_delta(dtype, q, k, scale=None):
    """Largest possible perturbation of a (natural-log) score by the operand rounding: eps * scale * max|q_i| * max|k_j|
    (one rounding of every c*q_d moves q_i.k_j by at most eps*scale*sum_d |q_d k_d| <= eps*scale*|q||k|)."""
    sc = q.shape[-1] ** -0.5 if scale is None else scale
    qn = float(np.sqrt((np.asarray(q, np.float64) ** 2).sum(-1)).max())
    kn = float(np.sqrt((np.asarray(k, np.float64) ** 2).sum(-1)).max())
    return PRESCALE_EPS[dtype] * sc * qn * kn


def lse_tol(dtype, prescaled, q, k, scale=None, base=1e-4):
    """LSE tolerance against the oracle on the TRUE Q: `base` (fp32 accumulation) plus, for the pre-scaled kernels, the bound
    the header states (LSE moves by at most the largest score perturbation)."""
    return base + rowsum_term(dtype, prescaled) + (prescale_delta(dtype, q, k, scale) if prescaled and dtype in PRESCALE_EPS else 0.0)


def o_tol(dtype, prescaled, q, k, v, scale=None, base=0.0):
    """O tolerance against the oracle on the TRUE Q: every probability moves by a factor within e^(+-delta), so
    |dO| <= 2 * (e^delta - 1) * max|v| on top of `base`."""
    if not prescaled or dtype not in PRESCALE_EPS:
        return base
    return base + 2.0 * float(np.expm1(prescale_delta(dtype, q, k, scale))) * float(np.abs(v).max())
