// fa_decode_kernel.hip -- few query rows against a long key sequence (decode steps, short chunks), entry point fa_fwd_decode.
//
// Scope row f3 (operator generality; not in the reference, whose operator is square: /root/reference/kernels.metal:606,619). The
// generalised forward fa_fwd_ex launches one workgroup per (batch, QUERY head, 128 query rows): a decode step -- 1..16 queries per
// head, grouped-query heads, thousands of keys -- then runs B.Hq workgroups on 256 CUs, each reading its key head's whole K and V,
// and streams 0.2-0.6 TB/s of an 8 TB/s memory (profiles/r03/decode_steps_splitkv_rule.log). This path is HBM-bound work and is
// laid out as such:
//   * the Hq / Hkv query heads that share a key/value head, times their Nq queries, are PACKED into the rows of one 16- or 32-row
//     query block (R = (Hq / Hkv) . Nq <= 32): K and V of a key head are read once, not once per query head;
//   * the keys of a (batch, key head) are SPLIT over S work items so that B . Hkv . S fills the chip several times over; an item is
//     ONE wave (a 64-thread workgroup: no barrier anywhere) that streams its keys in 64-key tiles through a private double buffer in
//     LDS by LDS-DMA (one tile in flight under the tile being multiplied), multiplies on v_mfma_f32_16x16x32 with the operand maps
//     and LDS images of fa_mfma16_kernel.hip, and keeps an exact online softmax (row maximum per tile: the arithmetic is free here);
//   * each item writes its partial (O . l, m, l) to a caller-owned workspace; a second launch combines the S partials of a row by
//     their maxima: M = max m_s, l = sum l_s 2^(m_s - M), O = sum O_s 2^(m_s - M) / l, LSE = (M + log2 l) ln 2.
// The library allocates nothing (C-ABI): the workspace size comes from fa_fwd_decode_workspace_bytes().
// e4m3 inputs (KV8; dtype FA_DTYPE_FP8_E4M3: Q, K, V e4m3 as in fa_fwd's config-5 family, O bf16): HALF the bytes of the stream this
// path is bound by. K and V tiles travel global -> registers (16 e4m3 per lane and load, the next tile's loads in flight under this
// tile's arithmetic) -> widened EXACTLY to bf16 -> the same swizzled LDS images, so everything behind the staging is the bf16 kernel
// bit for bit (results equal those of the bf16 path on the widened tensors). The widening is VALU work this path has to spare.
#include <stdlib.h>

#include <algorithm>

#include "fa_mfma_common.h"

#ifndef FA_DECODE_REGSTAGE
#define FA_DECODE_REGSTAGE 0  // 1 (experiment): 16-bit inputs also travel through registers into ONE LDS image (twice the items per CU), as e4m3 inputs do
#endif
#ifndef FA_DECODE_KV8_DEPTH
#define FA_DECODE_KV8_DEPTH 1  // e4m3 inputs, head_dim 64: tiles waiting in registers (1; 2 measured 0-13 % slower: 160 instead of 128 registers, profiles/r04/decode_ab_kv8_depth.log)
#endif

namespace fa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename Tag> struct MD16;
template <> struct MD16<BF16> {
  using elem = __bf16;
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MD16<F16> {
  using elem = _Float16;
  using vec8 = f16x8;
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// item (b, hkv, s): rows r = gi * Nq + iq of the packed block (gi: query head within the group), keys of tiles [t0, t1)
// 8 e4m3 (two dwords) -> 8 bf16 (four dwords = one 16-byte chunk): exact (bf16 = the upper half of the fp32 pattern)
__device__ __forceinline__ u32x4 widen8(unsigned w0, unsigned w1) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  unsigned r[4];
  const unsigned w[2] = {w0, w1};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[j], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[j], true);
    const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];  // (scalar temporaries: bit_cast on a vector element expression reads element 0)
    r[2 * j] = (__builtin_bit_cast(unsigned, a0) >> 16) | (__builtin_bit_cast(unsigned, a1) & 0xffff0000u);
    r[2 * j + 1] = (__builtin_bit_cast(unsigned, b0) >> 16) | (__builtin_bit_cast(unsigned, b1) & 0xffff0000u);
  }
  return u32x4{r[0], r[1], r[2], r[3]};
}

template <typename Tag, int D, int QT, bool CAUSAL, bool KV8>
__global__ __launch_bounds__(64) void decode_partial_kernel(DecodeParams p) {
  using M = MD16<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  constexpr int RB = D * 2;       // row bytes
  constexpr int CPR = D / 8;      // 16-byte chunks per row
  constexpr int KS = D / 32;      // 32-wide k-steps of the score product
  constexpr int DT = D / 16;      // 16-wide d tiles of O^T
  constexpr int KT = BN / 16;     // 16-key tiles per 64-key tile
  constexpr int TILE = BN * RB;   // bytes of one K (or V) tile
  constexpr int RPP = 1024 / RB;  // rows per 1-KiB LDS-DMA piece
  constexpr int NP = BN / RPP;    // pieces per tile and operand (all moved by this one wave)

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  constexpr bool RS = KV8 || (FA_DECODE_REGSTAGE != 0);  // tiles staged through registers into one LDS image
  lds_char *Kbuf = smem, *Vbuf = smem + (RS ? 1 : 2) * TILE;

  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const int S = p.S, item = blockIdx.x;
  const int bkv = item / S, s = item - bkv * S;
  const int b = bkv / p.Hkv, hkv = bkv - b * p.Hkv;
  const int G = p.Hq / p.Hkv, R = G * p.Nq;
  const int coff = p.Nk - p.Nq;
  const int nT = (p.Nk + BN - 1) / BN;
  const int t0 = (int)((long long)s * nT / S), t1 = (int)((long long)(s + 1) * nT / S);

  const long long base_kv = (long long)b * p.kv_bs + (long long)hkv * p.kv_hs;
  constexpr int EB = KV8 ? 1 : 2;  // bytes per input element
  const unsigned kv_bytes = (unsigned)p.Nk * D * EB;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.k + base_kv * EB), 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.v + base_kv * EB), 0, kv_bytes, 0x00020000);

  // ---- Q fragments: lane (c, g) holds row r = 16qt + c of the packed block, elements 32ks + 8g .. +7; pre-scaled Q~ = round(c.Q)
  const float c2 = p.scale * 1.4426950408889634f;
  vec8 qf[QT][KS];
  int rlim[QT];  // last visible key of the lane's rows (causal; Nk - 1 otherwise and for padding rows)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int r = 16 * qt + c;
    const int gi = r / p.Nq, iq = r - gi * p.Nq;
    const bool valid = r < R;
    rlim[qt] = (CAUSAL && valid) ? iq + coff : p.Nk - 1;
    const long long qoff = (long long)b * p.q_bs + (long long)(hkv * G + gi) * p.q_hs + (long long)iq * D;
    const elem *qp = (const elem *)p.q + qoff;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 t = {0u, 0u, 0u, 0u};
      if (KV8 && p.q8) {  // e4m3 queries (8 bytes), widened exactly; else (an e4m3 KV cache under 16-bit queries) as the 16-bit path
        if (valid) {
          const u32x2 t8 = *reinterpret_cast<const u32x2 *>((const char *)p.q + qoff + 32 * ks + 8 * g);
          const unsigned w0 = t8[0], w1 = t8[1];
          t = widen8(w0, w1);
        }
      } else {
        if (valid) t = *reinterpret_cast<const u32x4 *>(qp + 32 * ks + 8 * g);
      }
      qf[qt][ks] = __builtin_bit_cast(vec8, t);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (elem)((float)qf[qt][ks][j] * c2);
    }
  }

  // ---- per-lane LDS read addresses (images of fa_mfma16_kernel.hip)
  const int kx = (D == 64) ? ((c >> 1) & 7) : c;
  const lds_char *kptr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kptr[ks] = Kbuf + c * RB + (((4 * ks + g) ^ kx) << 4);
  const int vq = c >> 2, vp = c & 3, vrow = 4 * g + vq;
  const int vx = (D == 64) ? (((vrow >> 1) & 3) << 1) : ((vrow & 7) << 1);
  const lds_char *vptr[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) vptr[dt] = Vbuf + vrow * RB + ((((2 * dt) ^ vx) + (vp >> 1)) << 4) + 8 * (vp & 1);

  // ---- LDS-DMA: this wave moves every piece of a tile; piece j holds rows RPP j .. RPP j + RPP - 1, the chunk swizzle sits on the
  // source address: K chunk ^ ((row >> 1) & 7) (head_dim 64) / row & 15 (128), V chunk ^ (((row >> 1) & 3) << 1) / ((row & 7) << 1) --
  // the part of the swizzle that depends on the piece is a compile-time XOR of the byte offset
  const int drow = lane / CPR, dpc = lane % CPR;
  const unsigned dma_k0 = (unsigned)(drow * RB + ((dpc ^ ((D == 64) ? ((drow >> 1) & 7) : (drow & 15))) << 4));
  const unsigned dma_v0 = (unsigned)(drow * RB + ((dpc ^ ((D == 64) ? (((drow >> 1) & 3) << 1) : ((drow & 7) << 1))) << 4));
  auto stage_dma = [&](int t, int buf) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const unsigned kxo = (D == 64) ? (unsigned)((j & 1) << 6) : (unsigned)((j & 3) << 6);  // ((row >> 1) & 7) gains 4 on odd pieces / (row & 15) gains 4 (j & 3)
      const unsigned vxo = (D == 64) ? 0u : (unsigned)((j & 1) << 7);                        // head_dim 128: (row & 7) gains 4 on odd pieces -> chunk ^ 8
      const unsigned soff = (unsigned)t * TILE + j * 1024;
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)Kbuf + buf * TILE + j * 1024;
      const unsigned lv = (unsigned)(__UINTPTR_TYPE__)Vbuf + buf * TILE + j * 1024;
      const unsigned ko = dma_k0 ^ kxo, vo = dma_v0 ^ vxo;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(ko), "s"(rk), "s"(soff) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lv), "v"(vo), "s"(rv), "s"(soff) : "memory");
    }
  };

  // ---- KV8 staging: tile t = BN x D bytes per operand = NL loads of 16 e4m3 per lane; load i covers row (64 i + lane) 16 / D, bf16
  // chunks 2 ((lane 16 / 8) % CPR') .. +1 -- written into the images above with their swizzles
  constexpr int NL = RS ? D * EB / 16 : 1;
  // DEPTH tiles wait in registers (one; the knob allows two at head_dim 64)
  constexpr int DEPTH = (KV8 && D == 64) ? FA_DECODE_KV8_DEPTH : 1;
  static_assert(EB == 1 || EB == 2, "");
  u32x4 kraw[DEPTH][NL], vraw[DEPTH][NL];
  auto load_raw = [&](int t, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const unsigned off = (unsigned)t * (BN * D * EB) + (unsigned)(i * 64 + lane) * 16;
      kraw[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0);
      vraw[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0);
    }
  };
  auto write_tile = [&](auto slotc) {  // registers of slot -> the (one) LDS image
    constexpr int slot = decltype(slotc)::value;
    constexpr int buf = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e0 = (i * 64 + lane) * (16 / EB), row = e0 / D, ch = (e0 % D) / 8;  // (e4m3: ch even, chunks ch and ch + 1)
      const int ksw = (D == 64) ? ((row >> 1) & 7) : (row & 15);
      const int vsw = (D == 64) ? (((row >> 1) & 3) << 1) : ((row & 7) << 1);
      if constexpr (KV8) {
        const unsigned k0 = kraw[slot][i][0], k1 = kraw[slot][i][1], k2 = kraw[slot][i][2], k3 = kraw[slot][i][3];
        const unsigned v0 = vraw[slot][i][0], v1 = vraw[slot][i][1], v2 = vraw[slot][i][2], v3 = vraw[slot][i][3];
        lds_write_b128(Kbuf + buf * TILE + row * RB + ((ch ^ ksw) << 4), widen8(k0, k1));
        lds_write_b128(Kbuf + buf * TILE + row * RB + (((ch + 1) ^ ksw) << 4), widen8(k2, k3));
        lds_write_b128(Vbuf + buf * TILE + row * RB + ((ch ^ vsw) << 4), widen8(v0, v1));
        lds_write_b128(Vbuf + buf * TILE + row * RB + (((ch + 1) ^ vsw) << 4), widen8(v2, v3));
      } else {
        lds_write_b128(Kbuf + buf * TILE + row * RB + ((ch ^ ksw) << 4), kraw[slot][i]);
        lds_write_b128(Vbuf + buf * TILE + row * RB + ((ch ^ vsw) << 4), vraw[slot][i]);
      }
    }
  };

  f32x4 oacc[DT][QT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int i = 0; i < 4; ++i) oacc[dt][qt][i] = 0.0f;
  float m[QT], l[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = -INFINITY;
    l[qt] = 0.0f;
  }

  // one tile: (KV8) the tile is in the LDS image, register slot `slot` -- which held it -- is free for tile t + DEPTH; slot + 1 holds tile t + 1
  auto step = [&](auto slotc, const int t) {
      constexpr int slot = decltype(slotc)::value;
      // (KV8: ONE LDS image -- the next tile waits in registers and overwrites it behind this tile's arithmetic -- so an item holds half
      // the LDS and twice as many items, i.e. loads, are in flight per CU: the path is bound by tiles in flight, not by bytes)
      const int buf = RS ? 0 : ((t - t0) & 1);
      if constexpr (RS) {
        if (t + DEPTH < t1) load_raw(t + DEPTH, slotc);  // flies under this and the next tile's arithmetic; widened and written behind it
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t has landed (this wave issued every piece of it: no barrier needed)
        if (t + 1 < t1) stage_dma(t + 1, buf ^ 1);         // the next tile streams under this tile's arithmetic
      }
      const int kv0 = t * BN;
      const unsigned bo = (unsigned)(buf * TILE);
      // ---- S^T = K.Q~^T (log2 units): s[kt][qt][i] = S[row 16qt + c][key kv0 + 16kt + 4g + i]
      f32x4 sc[KT][QT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int i = 0; i < 4; ++i) sc[kt][qt][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const vec8 kf = __builtin_bit_cast(vec8, lds_read_b128(kptr[ks] + bo + kt * 16 * RB));
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) sc[kt][qt] = M::mfma(kf, qf[qt][ks], sc[kt][qt]);
        }
      }
      // ---- mask: key > the row's limit (causal, bottom-right aligned) or key >= Nk
      if (kv0 + BN > p.Nk || (CAUSAL && kv0 + BN - 1 > coff)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int lim = min(rlim[qt], p.Nk - 1) - kv0 - 4 * g;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[kt][qt][i] = (16 * kt + i > lim) ? -INFINITY : sc[kt][qt][i];
        }
      }
      // ---- exact online softmax (a row whose keys are all masked so far keeps m = -inf, l = 0: the guard avoids inf - inf)
      vec8 pf[2][QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = sc[0][qt][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) mx = fmaxf(mx, sc[kt][qt][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m[qt], mx);
        const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m[qt] - m_use);
        m[qt] = m_new;
        float ls = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sc[kt][qt][i] = __builtin_amdgcn_exp2f(sc[kt][qt][i] - m_use);
            ls += sc[kt][qt][i];
          }
        l[qt] = l[qt] * alpha + ls;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 4; ++i) oacc[dt][qt][i] *= alpha;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kp][qt][j] = (elem)sc[2 * kp + (j >> 2)][qt][j & 3];
      }
      // ---- O^T += V^T.P^T
#pragma unroll
      for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const lds_char *vb = vptr[dt] + bo + (32 * kp) * RB;
          const s16x4 lo = lds_read_tr16(vb), hi = lds_read_tr16(vb + 16 * RB);
          const s16x8 v8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = M::mfma(__builtin_bit_cast(vec8, v8), pf[kp][qt], oacc[dt][qt]);
        }
      if constexpr (RS) {
        if (t + 1 < t1) write_tile(std::integral_constant<int, (slot + 1) % DEPTH>{});  // (one wave, LDS operations in order: this tile's reads are behind us)
      }
  };
  if (t0 < t1) {
    if constexpr (RS) {
      load_raw(t0, std::integral_constant<int, 0>{});
      if constexpr (DEPTH == 2) {
        if (t0 + 1 < t1) load_raw(t0 + 1, std::integral_constant<int, 1>{});
      }
      write_tile(std::integral_constant<int, 0>{});
    } else {
      stage_dma(t0, 0);
    }
    for (int t = t0; t < t1; t += DEPTH) {
      step(std::integral_constant<int, 0>{}, t);
      if constexpr (DEPTH == 2) {
        if (t + 1 < t1) step(std::integral_constant<int, 1>{}, t + 1);
      }
    }
  }

  // ---- partial results -> workspace: [16 QT][D] O (unnormalised, fp32), then [16 QT][2] (m, l)
  float *wo = p.ws + (size_t)item * (16 * QT) * (D + 2);
  float *wm = wo + (16 * QT) * D;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const int row = 16 * qt + c;
    if (g == 0) {
      wm[2 * row] = m[qt];
      wm[2 * row + 1] = lt;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4 *>(wo + row * D + 16 * dt + 4 * g) = oacc[dt][qt];
  }
}

// one wave per output row (b, hq, iq). The S partials of a row are independent reads: lanes take the splits' (m, l) pairs in
// parallel (wave reductions for M and l), the weights 2^(m_s - M) go through LDS, and thread d folds element d (and d + 64) of the S
// partial rows with eight loads in flight (a serial loop over the splits -- one dependent ~0.4 us read each -- cost more than the
// streaming kernel itself: 47 us at S = 128).
template <typename Tag, int D, int QT>
__global__ __launch_bounds__(64) void decode_combine_kernel(DecodeParams p) {
  using elem = typename MD16<Tag>::elem;
  __shared__ float wgt[256];
  const int lane = threadIdx.x;
  const int row_id = blockIdx.x;  // (b * Hq + hq) * Nq + iq
  const int iq = row_id % p.Nq, bh = row_id / p.Nq;
  const int hq = bh % p.Hq, b = bh / p.Hq;
  const int G = p.Hq / p.Hkv, hkv = hq / G, gi = hq - hkv * G;
  const int r = gi * p.Nq + iq;
  const int S = p.S;  // <= 256
  constexpr int IST = (16 * QT) * (D + 2);
  const float *w0 = p.ws + (size_t)(b * p.Hkv + hkv) * S * IST;
  const float *ml = w0 + (16 * QT) * D + 2 * r;
  float ms[4], ls[4];
  float M = -INFINITY;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int s = lane + 64 * e;
    ms[e] = s < S ? ml[(size_t)s * IST] : -INFINITY;
    ls[e] = s < S ? ml[(size_t)s * IST + 1] : 0.0f;
    M = fmaxf(M, ms[e]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  float lsum = 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float f = (ms[e] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(ms[e] - M);  // a split that saw no visible key weighs nothing
    wgt[lane + 64 * e] = f;
    lsum += ls[e] * f;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  __syncthreads();
  float acc[D / 64];
#pragma unroll
  for (int e = 0; e < D / 64; ++e) acc[e] = 0.0f;
  const float *wr = w0 + r * D + lane;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float x[8][D / 64];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < D / 64; ++e) x[u][e] = wr[(size_t)(s + u) * IST + 64 * e];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < D / 64; ++e) acc[e] += x[u][e] * wgt[s + u];
  }
  for (; s < S; ++s)
#pragma unroll
    for (int e = 0; e < D / 64; ++e) acc[e] += wr[(size_t)s * IST + 64 * e] * wgt[s];
  const float inv = 1.0f / lsum;
  elem *op = (elem *)p.o + (long long)b * p.q_bs + (long long)hq * p.q_hs + (long long)iq * D;
#pragma unroll
  for (int e = 0; e < D / 64; ++e) op[lane + 64 * e] = (elem)(acc[e] * inv);
  if (p.lse != nullptr && lane == 0) p.lse[row_id] = (M + log2f(lsum)) * 0.6931471805599453f;
}

// ---------------------------------------------------------------------------
bool decode_supported(int dtype, int D) {
  return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16 || dtype == FA_DTYPE_FP8_E4M3) && (D == 64 || D == 128);
}

// splits per (batch, key head): one round of the chip's item slots (an item keeps one tile in flight: 4-5 / 2 items of 16 / 32 KiB per
// CU are what fills the memory pipe), but at least FA_DECODE_MIN_TILES 64-key tiles per item so that its prologue amortises
#ifndef FA_DECODE_MIN_TILES
#define FA_DECODE_MIN_TILES 4  // (2: -20 % at 8 key heads x 16384 keys; 8: +9 % at 32 x 16384 but -25 % at 4096; profiles/r04/decode_ab_split_heuristic.log)
#endif
#ifndef FA_DECODE_ROUNDS
#define FA_DECODE_ROUNDS 1
#endif
int decode_splits(int B, int Hkv, int Nk, int D, int kv8) {
  const int nT = (Nk + BN - 1) / BN;
  const int per_cu = ((D == 64) ? 5 : 2) * ((kv8 || FA_DECODE_REGSTAGE) ? 2 : 1);  // items resident per CU (32 / 64 KiB of LDS each; e4m3 inputs: one image, half)
  const long long want = (long long)FA_DECODE_ROUNDS * 256 * per_cu;
  long long S = (want + (long long)B * Hkv - 1) / ((long long)B * Hkv);
  S = std::min<long long>(S, std::max(1, nT / FA_DECODE_MIN_TILES));
  return (int)std::max<long long>(1, std::min<long long>(S, 256));
}

long long decode_workspace_bytes(int B, int Hq, int Hkv, int Nq, int Nk, int D) {
  const int R = (Hq / Hkv) * Nq, QT = (R + 15) / 16;
  // (sized for the larger of the two split counts -- e4m3 inputs run twice the items -- so that one workspace serves every dtype)
  return (long long)B * Hkv * std::max(decode_splits(B, Hkv, Nk, D, 0), decode_splits(B, Hkv, Nk, D, 1)) * (16 * QT) * (D + 2) * 4;
}

template <typename Tag, int D, int QT, bool KV8 = false>
static hipError_t launch_decode_q(const DecodeParams &p, hipStream_t s) {
  const size_t smem = ((KV8 || FA_DECODE_REGSTAGE) ? 2 : 4) * (size_t)BN * D * 2;
  (void)hipGetLastError();
  if (p.is_causal) {
    auto kern = decode_partial_kernel<Tag, D, QT, true, KV8>;
    if (smem > 48 * 1024) { hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3(p.B * p.Hkv * p.S), dim3(64), smem, s, p);
  } else {
    auto kern = decode_partial_kernel<Tag, D, QT, false, KV8>;
    if (smem > 48 * 1024) { hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3(p.B * p.Hkv * p.S), dim3(64), smem, s, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((decode_combine_kernel<Tag, D, QT>), dim3(p.B * p.Hq * p.Nq), dim3(64), 0, s, p);
  return hipGetLastError();
}

// dtype: of the queries (and of K / V unless kv8); kv8: K and V are e4m3 (dtype FP8: so are the queries; BF16: an e4m3 cache under
// bf16 queries -- the arithmetic is bf16 either way)
hipError_t launch_decode(const DecodeParams &p0, int D, int dtype, int kv8, hipStream_t s) {
  DecodeParams p = p0;
  p.q8 = (dtype == FA_DTYPE_FP8_E4M3);
  const int R = (p.Hq / p.Hkv) * p.Nq, QT = (R + 15) / 16;
  auto go = [&](auto tag) -> hipError_t {
    using Tag = decltype(tag);
    if (D == 64) return QT == 1 ? launch_decode_q<Tag, 64, 1>(p, s) : launch_decode_q<Tag, 64, 2>(p, s);
    if (D == 128) return QT == 1 ? launch_decode_q<Tag, 128, 1>(p, s) : launch_decode_q<Tag, 128, 2>(p, s);
    return hipErrorInvalidValue;
  };
  if (kv8 || dtype == FA_DTYPE_FP8_E4M3) {  // e4m3 K, V (and Q, or bf16 Q); bf16 arithmetic and output
    if (D == 64) return QT == 1 ? launch_decode_q<BF16, 64, 1, true>(p, s) : launch_decode_q<BF16, 64, 2, true>(p, s);
    if (D == 128) return QT == 1 ? launch_decode_q<BF16, 128, 1, true>(p, s) : launch_decode_q<BF16, 128, 2, true>(p, s);
    return hipErrorInvalidValue;
  }
  return dtype == FA_DTYPE_F16 ? go(F16{}) : go(BF16{});
}

}  // namespace fa
