/*
 * fa_mi355.h -- C-ABI of the MI355X (gfx950) attention-forward kernel library.
 *
 * This is the drop-in boundary for the reference's one operator,
 *     (Q, K, V, is_causal) -> O (+ LSE)
 * The reference has no FFI: its operator interface is the Metal binding table
 * of flash_attention_v4_half_kernel (/root/reference/kernels.metal:600-613) as
 * bound by the host at /root/reference/main.mm:821-852. fa_fwd() replaces that
 * table one argument for one slot:
 *
 *   buffer 0-2  Q,K,V  kernels.metal:601-603 / main.mm:822-824  -> q, k, v
 *   buffer 3    O      kernels.metal:604     / main.mm:825      -> o
 *   bytes  4    N      kernels.metal:605     / main.mm:826      -> N
 *   bytes  5    D      kernels.metal:606     / main.mm:827      -> D
 *   bytes  6    scale  kernels.metal:607     / main.mm:828      -> scale
 *   bytes  7,8  batch_stride, head_stride (elements)
 *                      kernels.metal:608-609 / main.mm:835-839  -> batch_stride, head_stride
 *   buffer 9    L_out  kernels.metal:611     / main.mm:840      -> lse  ([B,H,N] fp32)
 *   bytes  10   is_causal kernels.metal:612  / main.mm:842-843  -> is_causal
 *   grid (ceil(N/16), H, B) main.mm:846-852                     -> B, H (grid is the library's business)
 *
 * The other kernels the reference host dispatches by name
 * (naive_attention_kernel kernels.metal:12, flash_attention_kernel :72,
 * flash_attention_v2_kernel :462, flash_attention_simd_kernel :177;
 * looked up at main.mm:69-95) are selected with `variant`.
 *
 * Conventions (mirroring main.mm): the caller owns every buffer, all pointers
 * are DEVICE pointers, the library allocates nothing and keeps no state, the
 * launch is asynchronous on `hip_stream` (hipStream_t, may be NULL = default
 * stream) and re-entrant across devices/streams. Errors are returned, never
 * exit()ed (main.mm:16-22 exits; a library must not).
 */
#ifndef FA_MI355_H
#define FA_MI355_H

#ifdef __cplusplus
extern "C" {
#endif

#define FA_MI355_VERSION 400 /* major*10000 + minor*100 + patch */

/* element type of Q, K, V and O */
enum fa_dtype {
  FA_DTYPE_F32 = 0,      /* naive / v1 / v2 variants (kernels.metal:12,72,462) */
  FA_DTYPE_F16 = 1,      /* the reference operator's type (kernels.metal:601) */
  FA_DTYPE_BF16 = 2,     /* BASELINE.json configs 3,4 */
  FA_DTYPE_FP8_E4M3 = 3  /* Q,K,V OCP e4m3fn, O bf16, fp32 accumulate (BASELINE.json config 5); MFMA variant only */
};

/* which kernel computes the operator */
enum fa_variant {
  FA_VARIANT_AUTO = 0,   /* fastest kernel that supports (dtype, D) */
  FA_VARIANT_NAIVE = 1,  /* one thread per query row, two passes   (kernels.metal:12-64)   */
  FA_VARIANT_TILED = 2,  /* LDS-tiled scalar "V1"                  (kernels.metal:72-171)  */
  FA_VARIANT_TILED_V2 = 3, /* 128-bit loads, double-buffered K/V "V2" (kernels.metal:462-596) */
  FA_VARIANT_MFMA = 4,   /* matrix-core kernel "V3/V4"             (kernels.metal:177,600): 128 query rows per workgroup */
  FA_VARIANT_MFMA_PP = 5, /* RETIRED in library version 400 (reserved: fa_supported() answers 0, fa_fwd FA_ERR_UNSUPPORTED). Rounds 2-3:
                            the paired-block pipeline (256 query rows per workgroup, one wave per SIMD, asm-owned accumulators); the
                            128-row kernels beat it on every shape since they stage by LDS-DMA. Source: tools/experiments/ */
  FA_VARIANT_MFMA_SPLITKV = 6, /* same operator for small grids: one 32-row query block per workgroup, its keys split over
                            2-8 waves and merged in LDS through the row LSE (short sequences / few heads) */
  FA_VARIANT_MFMA_SPLIT2 = 7, /* same operator for grids that fill part of the chip: the 128-row workgroup of MFMA with eight
                            waves, waves 0-3 / 4-7 taking the even / odd KV tiles and merging once (halves the sequential
                            tile count of a block; head_dim 64, 128) */
  FA_VARIANT_MFMA_EXACT = 8, /* FA_VARIANT_MFMA without the pre-scaled query operand: every score is scaled in fp32 (one more
                            FMA per score, 6-8 % slower at head_dim 64). For logits far larger than a trained model
                            produces; see "LSE accuracy" below. Identical to FA_VARIANT_MFMA for fp8 inputs and D = 256 */
  FA_VARIANT_MFMA_H64S2 = 9, /* same operator, 64 query rows per workgroup: four waves, the two wave pairs take the even / odd
                            KV tiles and merge once (twice the workgroups, half the sequential tiles of a block): grids
                            whose critical path is the heaviest causal block (f16 / bf16, head_dim 64) */
  FA_VARIANT_MFMA16 = 10, /* FA_VARIANT_MFMA's workgroup (128 query rows, pre-scaled query operand) with every product on
                            v_mfma_f32_16x16x32 instead of 32x32x16: the chip holds a higher clock on that shape under
                            power-limited loops (f16 / bf16, head_dim 64) */
  FA_VARIANT_MFMA_FP8PV = 11 /* fp8 inputs, head_dim 64 or 128: BOTH products on the scaled fp8 MFMA -- the probabilities are rounded to e4m3
                            for the PV product (FA_VARIANT_MFMA / _EXACT keep them in bf16); see "fp8 probabilities" below */
};

/* status codes (0 = success, negative = error; text via fa_last_error()) */
enum fa_status {
  FA_OK = 0,
  FA_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad enum, misaligned */
  FA_ERR_UNSUPPORTED = -2,   /* (dtype, variant, D) combination has no kernel */
  FA_ERR_LAUNCH = -3,        /* HIP runtime reported an error at launch */
  FA_ERR_NO_DEVICE = -4      /* main.mm:42-45: no device */
};

/*
 * The operator. O[b,h,i,:] = sum_j softmax_j(scale * Q[b,h,i,:].K[b,h,j,:]) V[b,h,j,:]
 * with key j visible to query i iff (!is_causal || j <= i)   (kernels.metal:748);
 * lse[b,h,i] = max_j(scale*s_ij) + ln(sum_j exp(scale*s_ij - max))  (kernels.metal:862-864),
 * natural exp/log. Accumulation is fp32 for every dtype.
 *
 *  q,k,v,o       device pointers; element (b,h,i,d) at b*batch_stride + h*head_stride + i*D + d
 *                (rows contiguous, row pitch D); 16-byte aligned, strides multiples of 8 elements (16 for fp8)
 *  lse           device pointer to B*H*N floats, contiguous [B,H,N]; may be NULL
 *  N             sequence length (queries == keys); any N >= 1
 *  D             head dim: 32, 64, 96, 128 or 256 for FA_VARIANT_MFMA (fp8 inputs: 64, 128, 256); any multiple of 8 up to 128 for
 *                FA_VARIANT_MFMA16 (f16 / bf16: 64 and 128 natively, the others on zero-padded rows of the next larger one -- what
 *                FA_VARIANT_AUTO takes for head dims such as 40, 72, 80, 112; a head must then stay below 2 GiB); 64 or 128 for
 *                FA_VARIANT_MFMA_SPLIT2 / _FP8PV, 64 for _H64S2 / _SPLITKV, <= 128 (multiple of 4) for the scalar variants
 *  scale         softmax scale (> 0); the reference passes 1/sqrt(D) (main.mm:13)
 *  dtype/variant enums above
 *  hip_stream    hipStream_t on the current device, or NULL
 * Inputs are expected to be finite: the matrix-core kernels are compiled without NaN handling (their
 * own -inf mask values never meet anything that could produce one), so NaN/Inf in Q, K or V give
 * unspecified output values (never a fault).
 *
 * LSE accuracy. The reference never checks L_out (main.mm:1083 only feeds it to its backward) and forms its scores in
 * half precision (kernels.metal:709-712). Here every sum is fp32. The kernels with a PRE-SCALED query operand -- FA_VARIANT_MFMA,
 * FA_VARIANT_MFMA_SPLIT2, FA_VARIANT_MFMA_H64S2 and FA_VARIANT_MFMA16 with f16 / bf16 inputs and D <= 128 (what FA_VARIANT_AUTO
 * picks for BASELINE configs 3 and 4, and the 128-row route of fa_fwd_ex) -- multiply the query operand by scale*log2(e) and round
 * it to the input type ONCE per block of query rows, so that the matrix core delivers the exponent of every probability directly
 * (replaces the per-score scale-and-subtract of kernels.metal:763-771): the result is the exact operator applied to a Q' with
 * |Q' - Q| <= eps*|Q| element-wise, eps = 2^-9 (bf16) / 2^-12 (f16), half an ulp of the input type. Consequently
 *     |lse - exact| <= 1e-4 + eps * scale * |q_i|_2 * max_j |k_j|_2        (row i),
 * i.e. relative to the score magnitude (measured on BASELINE config 3, U(-1,1) inputs: max 1.0e-3, rms 4e-5 for
 * bf16; 1.3e-4 / 5e-6 for f16), while O keeps the tolerance of the other kernels (max|O - exact| 3.1e-3 vs 2.9e-3
 * without the pre-scaling on config 3). FA_VARIANT_MFMA16 additionally takes its row sums from the matrix core, i.e. it adds the
 * probabilities AFTER their rounding to the input type (the very values the PV product multiplies: O's weights then add up to
 * exactly 1): ln(l) carries that rounding, at most 2^-8 (bf16) / 2^-11 (f16) on top of the bound above and far less on average
 * (measured: 4e-4 at N = 128, below 1e-4 from N = 1024 on, bf16). Every other kernel / dtype (FA_VARIANT_MFMA_EXACT, the split-KV
 * and paired-block kernels, fp8 inputs): |lse - exact| <= 1e-4 for |lse| <= ~10.
 *
 * fp8 probabilities. FA_VARIANT_MFMA_FP8PV (e4m3 inputs, D = 64 or 128; FA_VARIANT_AUTO's choice for grids that fill the chip) runs BOTH
 * products on the fp8 matrix pipe: the probabilities are rounded to e4m3 (3 mantissa bits) on their way into the PV product, as the
 * inputs themselves were. Every softmax weight moves by a factor within 1 +- 2^-4, so
 *     |O - exact| <= 2^-4 * max_j |v_j|     (plus the bf16 rounding of O),
 * reached only by rows with one or two visible keys; the roundings of a row's weights are independent, so rows with many comparable
 * keys see a small fraction of it (measured on BASELINE config 5, U(-1,1) inputs: max 2.3e-2 over the first rows of the causal mask,
 * below the other kernels' 6e-3 on every row with more than 1024 keys). Weights below 2^-13 of the row's reference are flushed to
 * zero (e4m3's range). The row sum comes out of the matrix core as well (a block of ones against the e4m3 probabilities): l adds the
 * ROUNDED weights -- O's weights add up to exactly 1 -- and
 *     |lse - exact| <= ln(1 + 2^-4) < 2^-4,
 * again reached only by rows with two or three comparable keys (measured on config 5: at most 4.1e-3 over rows with more than 64
 * keys, 1e-3 typical). Callers that need the 1e-4 LSE with fp8 inputs name FA_VARIANT_MFMA / FA_VARIANT_MFMA_EXACT, which keep the
 * probabilities in bf16 and add them in fp32 (the score product alone on the fp8 pipe).
 */
int fa_fwd(const void *q, const void *k, const void *v, void *o, float *lse,
           int B, int H, int N, int D, float scale,
           long long batch_stride, long long head_stride,
           int is_causal, int dtype, int variant, void *hip_stream);

/*
 * Generalised forward (scope row f3; not in the reference, whose operator is square and multi-head):
 * grouped-query / multi-query heads and a key length different from the query length.
 *   q, o  [B, Hq, Nq, D] with q_batch_stride / q_head_stride;  k, v  [B, Hkv, Nk, D] with kv strides
 *   query head h attends to key/value head h / (Hq / Hkv)   (Hq % Hkv == 0)
 *   causal is bottom-right aligned: key j is visible to query i iff j <= i + (Nk - Nq); needs Nk >= Nq
 *   lse [B, Hq, Nq]. Matrix-core kernels only (f16 / bf16: D = 32, 64, 96, 128, 256; fp8 inputs: D = 64, 128, 256): the 128-row
 *   kernel, or -- at most 64 blocks of 128 query rows against more than 64 keys, e.g. decode steps, D = 64 -- the
 *   split-KV kernel, as FA_VARIANT_AUTO chooses for fa_fwd.
 * With Hkv = Hq, Nk = Nq and equal strides this is fa_fwd(..., FA_VARIANT_MFMA) (or ..._SPLITKV under that rule).
 */
int fa_fwd_ex(const void *q, const void *k, const void *v, void *o, float *lse,
              int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
              long long q_batch_stride, long long q_head_stride,
              long long kv_batch_stride, long long kv_head_stride,
              int is_causal, int dtype, void *hip_stream);
/* fa_fwd_ex with the kernel named by the caller: FA_VARIANT_AUTO (= fa_fwd_ex), FA_VARIANT_MFMA, FA_VARIANT_MFMA_EXACT (no pre-scaled
 * query operand), FA_VARIANT_MFMA16 (f16 / bf16, D = 64, 128) or FA_VARIANT_MFMA_SPLITKV (D = 64); any other variant: FA_ERR_UNSUPPORTED. */
int fa_fwd_exv(const void *q, const void *k, const void *v, void *o, float *lse,
               int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
               long long q_batch_stride, long long q_head_stride,
               long long kv_batch_stride, long long kv_head_stride,
               int is_causal, int dtype, int variant, void *hip_stream);

/*
 * Few query rows against a long key sequence (decode steps, short chunks; scope row f3, not in the reference): the same operator as
 * fa_fwd_ex for (Hq / Hkv) * Nq <= 32, f16 / bf16 / e4m3, D = 64 | 128, laid out for the HBM roofline instead of the matrix cores -- the query
 * heads of a key/value head are packed into one row block (K and V are read once per key head), the keys are split over several
 * work items per (batch, key head), and the partial results (unnormalised O, m, l per item) meet in `workspace`, caller-owned device
 * memory of fa_fwd_decode_workspace_bytes() bytes, 16-byte aligned, contents irrelevant before and after the call; a second launch
 * on the same stream combines them. Pre-scaled query operand as FA_VARIANT_MFMA ("LSE accuracy" above). Asynchronous, allocates nothing.
 * dtype FA_DTYPE_FP8_E4M3 (an e4m3 KV cache: Q, K, V e4m3 under strides that are multiples of 16, O bf16 as in fa_fwd): half the bytes of
 * the stream; the tiles are widened EXACTLY to bf16 on their way into LDS, so the arithmetic and the tolerances are the bf16 path's (the
 * probabilities stay bf16 here). The workspace size does not depend on the dtype.
 */
int fa_fwd_decode(const void *q, const void *k, const void *v, void *o, float *lse,
                  int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
                  long long q_batch_stride, long long q_head_stride,
                  long long kv_batch_stride, long long kv_head_stride,
                  int is_causal, int dtype, void *workspace, long long workspace_bytes, void *hip_stream);
/* The same step on an e4m3 KV cache under 16-bit queries, the usual serving layout: k, v e4m3 (kv strides in elements, multiples of 16),
 * q and o bf16 (q_dtype FA_DTYPE_BF16; FA_DTYPE_FP8_E4M3 makes this fa_fwd_decode with dtype e4m3). K and V are widened exactly to bf16
 * on their way into LDS: the arithmetic and tolerances are those of the bf16 path on the widened cache. Same workspace. */
int fa_fwd_decode_kv8(const void *q, const void *k, const void *v, void *o, float *lse,
                      int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
                      long long q_batch_stride, long long q_head_stride,
                      long long kv_batch_stride, long long kv_head_stride,
                      int is_causal, int q_dtype, void *workspace, long long workspace_bytes, void *hip_stream);
long long fa_fwd_decode_workspace_bytes(int B, int Hq, int Hkv, int Nq, int Nk, int D);
int fa_fwd_decode_supported(int dtype, int D, int Hq, int Hkv, int Nq);

/*
 * Backward of the operator (row f1 of the scope table): the reference binds it as
 * flash_attention_backward_kernel, /root/reference/kernels.metal:905-921, host side
 * /root/reference/main.mm:1015-1058:
 *   buffer 0-5  Q,K,V,O,dO (16-bit), L (fp32 [B,H,N], the forward's lse)   -> q,k,v,o,d_o,lse
 *   buffer 6-8  dQ,dK,dV  fp32 (the reference accumulates into them with atomics and the host
 *               zero-fills them first, main.mm:1018-1021)                  -> dq,dk,dv
 *   bytes 9-14  N, D, scale, batch_stride, head_stride, is_causal          -> same names
 * Here dq/dk/dv are WRITTEN (no zero-fill needed, no atomics, bitwise reproducible), laid out
 * like the inputs (element (b,h,i,d) at b*batch_stride + h*head_stride + i*D + d, fp32).
 * `workspace` is caller-owned scratch of fa_bwd_workspace_bytes(B,H,N) bytes (device memory).
 * dtype F16 or BF16 (FP8_E4M3: see fa_bwd_workspace_bytes_ex); D = 64 (the reference's, kernels.metal:905-1265) or 128 natively, any other multiple of 8 up to 128 (32, 96, ...)
 * through the next larger kernel on zero-padded rows (same results per real column; a head must then stay below 2 GiB). D = 256: its own
 * instantiation (one workgroup per CU, f16 / bf16 only).
 */
int fa_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse,
           float *dq, float *dk, float *dv, void *workspace,
           int B, int H, int N, int D, float scale,
           long long batch_stride, long long head_stride,
           int is_causal, int dtype, void *hip_stream);
/* Generalised backward, the counterpart of fa_fwd_ex (scope row f3 applied to f1; the reference's operator is square and has one
 * head count, kernels.metal:906-920): q, o, d_o, dq are [B,Hq,Nq,D] under the q strides, k, v, dk, dv are [B,Hkv,Nk,D] under the
 * kv strides, Hq % Hkv == 0, query head h reads key/value head h / (Hq / Hkv); causal is bottom-right aligned (key j visible to
 * query i iff j <= i + Nk - Nq; needs Nk >= Nq) exactly as in fa_fwd_ex. dK / dV of a key head are the sums over its Hq / Hkv
 * query heads, formed in registers by the one workgroup that owns the key block (no atomics, bitwise reproducible). lse is
 * [B,Hq,Nq]; workspace is fa_bwd_workspace_bytes(B,Hq,Nq). With Hkv = Hq, Nk = Nq and equal strides this is exactly fa_bwd. */
int fa_bwd_ex(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse,
              float *dq, float *dk, float *dv, void *workspace,
              int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
              long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
              int is_causal, int dtype, void *hip_stream);
long long fa_bwd_workspace_bytes(int B, int H, int N);
/* the same for any dtype and for fa_bwd_ex's shapes (fa_bwd: Hq = Hkv, Nq = Nk, one stride pair). dtype FA_DTYPE_FP8_E4M3 (Q, K, V
 * e4m3 under their element strides, multiples of 16; O and d_o bf16 -- what fa_fwd writes for e4m3 inputs -- under the same element
 * strides; D a multiple of 16): the backward first widens Q, K, V to bf16 -- exactly -- into this workspace, which is why it grows by two
 * bytes per element of their extents, and then runs the bf16 kernels. For f16 / bf16 it returns fa_bwd_workspace_bytes(B,Hq,Nq) rounded up. */
long long fa_bwd_workspace_bytes_ex(int dtype, int B, int Hq, int Hkv, int Nq, int Nk, int D, long long q_batch_stride,
                                    long long q_head_stride, long long kv_batch_stride, long long kv_head_stride);
int fa_bwd_supported(int dtype, int D);
/* algorithmic FLOPs of one fa_bwd call: 2.5x the forward (five N x N x D products) */
double fa_bwd_algorithmic_flops(int B, int H, int N, int D, int is_causal);

/* 1 if fa_fwd has a kernel for the combination, else 0 (no GPU needed). */
int fa_supported(int dtype, int variant, int D);

/* variant FA_VARIANT_AUTO resolves to for (dtype, D); FA_ERR_UNSUPPORTED if none. */
int fa_resolve_variant(int dtype, int D);

/* variant FA_VARIANT_AUTO resolves to for a whole problem (the choice between the matrix-core kernels depends on
 * the grid the shape gives); FA_ERR_UNSUPPORTED if none. */
int fa_resolve_variant_for(int dtype, int D, int B, int H, int N, int is_causal);

/* name of the device kernel fa_fwd(..., FA_VARIANT_AUTO) launches for the problem, as rocprofv3 prints it
 * (e.g. "fa::fwd_pp_kernel<fa::BF16, 64, true>"); "" if none. Static storage. */
const char *fa_fwd_kernel_name(int dtype, int D, int B, int H, int N, int is_causal);

/* bytes per element of Q/K/V and of O for a dtype (fp8: 1 and 2). 0 if bad enum. */
int fa_dtype_in_bytes(int dtype);
int fa_dtype_out_bytes(int dtype);

/* algorithmic work of one fa_fwd call:
 * flops = 4*B*H*N^2*D (2*.. causal); bytes = (3*in + out)*B*H*N*D + 4*B*H*N */
double fa_algorithmic_flops(int B, int H, int N, int D, int is_causal);
double fa_algorithmic_bytes(int B, int H, int N, int D, int dtype);

/* thread-local text of the last error on this thread ("" if none). */
const char *fa_last_error(void);

int fa_version(void);
const char *fa_variant_name(int variant);
const char *fa_dtype_name(int dtype);

#ifdef __cplusplus
}
#endif
#endif /* FA_MI355_H */
