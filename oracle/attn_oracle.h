/*
 * attn_oracle.h -- CPU oracle for the attention-forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it,
 * and only as the checker / the timed CPU baseline. The product path
 * (flash_attention_metal_amd/csrc, driver/) never links or loads this library.
 *
 * It is a plain-C restatement of the reference's inline CPU checks:
 *   /root/reference/main.mm:24-30    initRandom (mt19937(42) -> U(-1,1))
 *   /root/reference/main.mm:128-159  non-causal fp32 oracle, O(N^2 D^2)
 *   /root/reference/main.mm:551-578  causal fp32 oracle, O(N^2 D)
 *   /root/reference/main.mm:13       SCALE = 1/sqrt(D)
 *   /root/reference/kernels.metal:862-864  L = m + log(l)   (no CPU counterpart
 *       in the reference: LSE parity is pinned by this file alone)
 *
 * Pinning: oracle/build_ref.sh compiles the reference's own oracle line ranges
 * (sliced by line number from /root/reference/main.mm at build time, never
 * copied into the repo) into oracle/_ref/libfa_ref_slices.so; tests/golden/
 * holds the outputs it produced (tests/golden/make_golden.py) and
 * tests/test_oracle.py checks this restatement against them bit for bit.
 */
#ifndef FA_ATTN_ORACLE_H
#define FA_ATTN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* main.mm:24-30 with a selectable seed (the reference always uses 42, so its
 * Q == K == V). libstdc++ semantics of uniform_real_distribution<float>. */
void oracle_init_random(float *data, long long size, uint32_t seed);

/* main.mm:128-159 verbatim loop structure: every output element (i,d)
 * recomputes all N scores twice. Single head, fp32. */
void oracle_noncausal_faithful(const float *q, const float *k, const float *v,
                               float *o, int N, int D, float scale);

/* Same arithmetic with the score rows hoisted out of the d loop (O(N^2 D)):
 * per output element the operation order is unchanged, so the result is
 * bit-identical to oracle_noncausal_faithful (tested). Also emits LSE. */
void oracle_noncausal_hoisted(const float *q, const float *k, const float *v,
                              float *o, float *lse /*nullable*/, int N, int D,
                              float scale);

/* main.mm:551-578 loop structure (scores j<=i, max, exp+sum, then PV). */
void oracle_causal(const float *q, const float *k, const float *v, float *o,
                   float *lse /*nullable*/, int N, int D, float scale);

/* The operator over [B,H,N,D] with element strides (kernels.metal:622):
 * non-causal heads use the hoisted form, causal heads the causal form.
 * threads <= 1 runs serially; otherwise OpenMP over (b,h,row-block). */
void oracle_attn_fwd(const float *q, const float *k, const float *v, float *o,
                     float *lse /*nullable*/, int B, int H, int N, int D,
                     float scale, long long batch_stride,
                     long long head_stride, int is_causal, int threads);

/* fp64 accumulation everywhere (inputs still fp32): the tight error anchor for
 * the 16-bit kernels. Outputs are double. */
void oracle_attn_fwd_f64(const float *q, const float *k, const float *v,
                         double *o, double *lse /*nullable*/, int B, int H,
                         int N, int D, float scale, long long batch_stride,
                         long long head_stride, int is_causal, int threads);

/* fp64 result for selected query rows of one contiguous [N,D] head. */
void oracle_attn_rows_f64(const float *q, const float *k, const float *v,
                          double *o /*[nrows,D]*/, double *lse /*[nrows], nullable*/,
                          int N, int D, float scale, int is_causal,
                          const int *rows, int nrows, int threads);

/* Generalised operator in fp64: grouped-query heads (Hq % Hkv == 0) and Nq != Nk with bottom-right
 * causal alignment (key j visible to query i iff j <= i + Nk - Nq). Contiguous q,o [B,Hq,Nq,D],
 * k,v [B,Hkv,Nk,D], lse [B,Hq,Nq]. Not in the reference: unpinned. */
void oracle_attn_fwd_ex_f64(const float *q, const float *k, const float *v, double *o, double *lse,
                            int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
                            int is_causal, int threads);

/* fp64 gradients of the operator over contiguous [B,H,N,D] (kernels.metal:905-1265 math).
 * Backward parity is unpinned by the reference (its CPU check is broken, main.mm:1100-1101). */
void oracle_attn_bwd_f64(const float *q, const float *k, const float *v, const float *d_o,
                         double *dq, double *dk, double *dv, int B, int H, int N, int D,
                         float scale, int is_causal, int threads);

/* Round-to-nearest-even casts used to build 16-bit / fp8 test inputs
 * (main.mm:322-329 does the fp16 one with a __fp16 cast). In place, fp32 -> T
 * -> fp32. fp8 is OCP e4m3fn with saturation to +-448 (NaN stays NaN). */
void oracle_round_f16(float *x, long long n);
void oracle_round_bf16(float *x, long long n);
void oracle_round_fp8_e4m3(float *x, long long n);
/* Bit-pattern converters (RNE) for feeding the device buffers. */
uint16_t oracle_f32_to_f16_bits(float x);
uint16_t oracle_f32_to_bf16_bits(float x);
uint8_t oracle_f32_to_fp8_e4m3_bits(float x);
float oracle_f16_bits_to_f32(uint16_t h);
float oracle_bf16_bits_to_f32(uint16_t h);
float oracle_fp8_e4m3_bits_to_f32(uint8_t b);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
