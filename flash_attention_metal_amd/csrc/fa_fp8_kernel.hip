// fa_fp8_kernel.hip -- the operator for OCP e4m3 inputs with BOTH products on the fp8 matrix pipe (variant "mfma_fp8pv"; head_dim 64, 128).
//
// BASELINE.json configs[4]: "fp8 Q/K/V with fp32 accumulate (CDNA4 fp8 MFMA)". Same math as the other matrix-core kernels (replaces
// /root/reference/kernels.metal:600-883; mask kernels.metal:748, L = m + ln(l) kernels.metal:862-864). fa_mfma_kernel.hip runs only
// the score product on v_mfma_scale_f32_32x32x64_f8f6f4 and widens V to bf16 on its way to LDS (a VGPR round trip, conversion VALU,
// twice the LDS bytes, PV at the bf16 rate). Here (one workgroup = 4 waves = 128 query rows, a wave owns 32):
//   * K AND V stay e4m3 in LDS (rows of D bytes) and travel global -> LDS by LDS-DMA, swizzle on the source address
//   * S^T = K.Q^T on the scaled fp8 MFMA with unit E8M0 scales (an exact e4m3 product, 64 head-dim elements per instruction)
//   * P is rounded to e4m3 (v_cvt_pk_fp8_f32) and O^T += V^T.P^T runs on the same instruction, 64 keys per MFMA: the B operand's k
//     index (32h + j) is key 32(j >> 4) + 8((j >> 2) & 3) + 4h + (j & 3) -- the order the scores sit in the accumulator -- and V^T is
//     fetched in that order by ds_read_b64_tr_b8 (per 16-lane group: 8 rows x 16 bytes, lane 2q + p supplies row q's bytes 8p..8p+7,
//     lane i receives column i with row q in byte q; probed on the hardware: tools/probes/probe_tr8.hip)
//   * e4m3 holds 2^-9 .. 448 with 3 mantissa bits: the probabilities are formed against a reference SHIFT = 3 (log2 units) BELOW the
//     row maximum (P' = 8 P <= 8 when the reference is set; O and l carry the same factor, it cancels), which leaves 2^-13 of the row
//     maximum above the flush-to-zero floor. The conversion does not saturate (480 -> NaN, probed), and that is the staleness test:
//   * the tile's row sums are a third MFMA, ones . P^T, into a fresh accumulator (every register of the tuple = the sum of row r's 64
//     rounded probabilities; the V^T fragments are fetched under it): a probability past 448 makes that sum NaN, the wave renews its
//     reference from the tile's maximum and repeats the tile (a backward branch, rare). No additions, no per-element test on the
//     vector pipe -- the kernel is VALU-bound (90 % busy, the matrix pipe 31 %): config 5 1457 -> 1585 TFLOP/s, interleaved
//     (profiles/r04/ab_fp8_ones_rowsums.log). FA8_ONES = 0 keeps the first form (31 additions per tile, renewal when a lane's 32
//     probabilities add up to more than 448).
//     The price of the e4m3 probabilities is their rounding to 3 mantissa bits, in O and now in l: include/fa_mi355.h, "fp8 probabilities".
// Head dim 128 (the same kernel, template D): rows of 128 bytes, two 64-deep k-steps per score tuple, four PV products + the ones product
// per tile, two LDS-DMA pieces per wave, tile and operand, its own chunk swizzles (two rows share a 64-bank line: K by (row >> 1) & 7;
// V by chunk pairs so that the eight rows of a transposed read spread over the banks), 166 registers = three workgroups per CU:
// 2 x 16 heads x 8192 causal 1410 (bf16 probabilities) -> 2021 TFLOP/s (profiles/r04/ab_fp8pv_head_dim_128.log).
#include <stdlib.h>

#include <algorithm>

#include "fa_mfma_common.h"

#ifndef FA8_SHIFT
#define FA8_SHIFT 3.0f  // log2 units: the reference sits this far BELOW the row maximum found when it was last set
#endif
#ifndef FA8_SUM_LIMIT
#define FA8_SUM_LIMIT 448.0f  // a lane's 32 probabilities of a tile may add up to this before the reference is renewed (e4m3 max)
#endif
#ifndef FA8_ONES
#define FA8_ONES 1  // 1: the row sums come out of the matrix core (a third "d block" of ones against the e4m3 probabilities: l adds the
#endif              // ROUNDED probabilities, the very weights of the PV product) and a stale reference shows as a NaN sum;
                    // 0: round 4's first form, 31 additions per tile and a sum limit
#ifndef FA8_OCC
#define FA8_OCC 4
#endif
#ifndef FA8_OCC128
#define FA8_OCC128 3  // workgroups per CU of the head_dim-128 instantiation (32 KiB of LDS; 168 registers)
#endif

namespace fa {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// the scaled fp8 MFMA with unit scales (E8M0 0x7f = 2^0): D = A(e4m3, 32 x 64) . B(e4m3, 64 x 32) + C
__device__ __forceinline__ f32x16 mfma_f8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int D, bool CAUSAL>
__device__ __forceinline__ void fwd_fp8_body(const Params &p) {
  static_assert(D == 64 || D == 128, "head dims of the all-fp8 kernel");
  constexpr int RW = BM / WM;     // waves
  constexpr int KS = D / 64;      // 64-deep k-steps of the score product
  constexpr int CPR = D / 16;     // 16-byte chunks of a K / V row
  constexpr int RB = D;           // row bytes of K / V (global and LDS)
  constexpr int ORB = 2 * D;      // row bytes of the bf16 O tile
  constexpr int TILE = BN * RB;   // bytes of one K (or V) tile
  constexpr int DB = D / 32;      // 32-wide d blocks of O^T
  constexpr int CPO = ORB / 16;   // 16-byte chunks of an O row

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  lds_char *Kbuf = smem;             // [2][BN][RB], chunks swizzled
  lds_char *Vbuf = smem + 2 * TILE;  // [2][BN][RB], chunks swizzled

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 31;  // query within the wave / key row within a 32-key block / d within a 32-wide block
  const int h = lane >> 5;  // lane half

  int bh, qb;
  map_block<CAUSAL>(blockIdx.x, p, bh, qb);
  long long base, base_kv;
  head_bases(bh, p, base, base_kv);
  const int coff = p.Nk - p.N;
  const int q0 = qb * BM;
  const int qw0 = q0 + wave * WM;
  const int qrow = qw0 + r;

  const unsigned head_bytes = (unsigned)p.N * RB, kv_head_bytes = (unsigned)p.Nk * RB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.q + base), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.k + base_kv), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.v + base_kv), 0, kv_head_bytes, 0x00020000);

  // ---- Q fragment (B operand of K.Q^T): lane (r, h) holds Q[qrow][32h .. 32h+31] (any k order works as long as K uses the same)
  // (head_dim 128: k-step ks covers columns 64ks .. 64ks+63)
  i32x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * RB + 64 * ks + 32 * h, 0, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * RB + 64 * ks + 32 * h + 16, 0, 0);
    qf[ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  }

  // ---- per-lane LDS addresses (absolute, opaque to hipcc)
  // K: row (32kb + r), bytes 32h .. 32h+31 = 16-byte chunks 2h, 2h+1, swizzled by (r >> 2) & 3 (conflict-free ds_read_b128)
  // (head_dim 128: rows of 128 bytes = 8 chunks, two rows per 64-bank line: chunks 4ks + 2h, 4ks + 2h + 1 swizzled by (r >> 1) & 7)
  auto k_swz = [](int row) { return D == 64 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
  const int kx = k_swz(r);
  const lds_char *kptr0[KS], *kptr1[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kptr0[ks] = Kbuf + r * RB + (((4 * ks + 2 * h) ^ kx) << 4);
    kptr1[ks] = Kbuf + r * RB + (((4 * ks + 2 * h + 1) ^ kx) << 4);
    asm volatile("" : "+v"(kptr0[ks]), "+v"(kptr1[ks]));
  }
  // V: ds_read_b64_tr_b8. 16-lane group (h, c16 = (lane >> 4) & 1) covers d columns 32db + 16c16 ..+15; its lane 2q + pp supplies the
  // address of row q of the group's 8 rows, bytes 8pp .. 8pp+7: rows q = 0..3 -> keys 8g + 4h + q, q = 4..7 -> keys 8(g+1) + 4h + (q-4)
  // (g = 0, 2 per read; + 32kb), so the two result dwords are elements j = 16kb + 4g + 0..3 and 16kb + 4(g+1) + 0..3 of the operand.
  // Image: 16-byte chunk index ^ (((row >> 3) & 1) << 1): the two 4-row halves of a read sit in different 32-byte halves of the bank row.
  // (head_dim 128: 128-byte rows, two per 64-bank line: the eight rows of a read -- keys 4h + 0..3 and 8 + 4h + 0..3 -- must spread
  // over four chunk PAIRS per row parity: chunk index ^ ((bit 1 of the row | bit 3 of the row << 1) << 1); the group's two chunks of
  // a 32-byte pair stay adjacent)
  auto v_swz = [](int row) { return D == 64 ? (((row >> 3) & 1) << 1) : ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1); };
  const int vi = lane & 15, vq = vi >> 1, vpp = vi & 1, c16 = (lane >> 4) & 1;
  const int vrow = 8 * (vq >> 2) + 4 * h + (vq & 3);
  const lds_char *vptr[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    vptr[db] = Vbuf + vrow * RB + (((2 * db + c16) ^ v_swz(vrow)) << 4) + 8 * vpp;
    asm volatile("" : "+v"(vptr[db]));
  }

  const int kv_end = CAUSAL ? min(p.Nk, q0 + BM + coff) : p.Nk;
  const int nT = (kv_end + BN - 1) / BN;

  // ---- LDS-DMA staging: one 1-KiB piece (16 rows) of K and one of V per wave and tile; the swizzles sit on the SOURCE address
  // (head_dim 128: two pieces of 8 rows per wave, tile and operand -- pieces w and w + 4, 32 rows apart: the swizzles repeat every 16)
  constexpr int RPP = 1024 / RB, NPW = (BN / RPP) / RW;  // rows per piece; pieces per wave, tile and operand
  static_assert((RW * RPP) % 16 == 0, "the piece stride must keep the swizzles");
  unsigned dma_ko, dma_vo;
  {
    const int row = wave * RPP + lane / CPR, pc = lane % CPR;
    dma_ko = (unsigned)(row * RB + ((pc ^ k_swz(row)) << 4));
    dma_vo = (unsigned)(row * RB + ((pc ^ v_swz(row)) << 4));
  }
  auto stage_dma = [&](int t, int buf) {  // tile t -> buffer buf (hipcc does not count these loads: the caller waits vmcnt(0))
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const unsigned soff = (unsigned)t * TILE + j * (RW * 1024);
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)Kbuf + buf * TILE + (wave + RW * j) * 1024;
      const unsigned lv = (unsigned)(__UINTPTR_TYPE__)Vbuf + buf * TILE + (wave + RW * j) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(dma_ko), "s"(rk), "s"(soff) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lv), "v"(dma_vo), "s"(rv), "s"(soff) : "memory");
    }
  };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.0f;
  float mref = -INFINITY;  // reference of this row, log2 units: (a stale) row maximum - SHIFT
  constexpr bool ONES = (FA8_ONES != 0);
  float l = 0.0f;          // !ONES: this lane half's share of the row sum (of the unrounded probabilities); ONES: the whole row's sum
  i32x8 ones8;             // e4m3 1.0 = 0x38 in every byte
#pragma unroll
  for (int i = 0; i < 8; ++i) ones8[i] = 0x38383838;
  if constexpr (ONES) asm volatile("" : "+v"(ones8));  // (kept in registers: else re-materialised in front of every use)
  const float c2 = p.scale * 1.4426950408889634f;  // scale * log2(e)

  stage_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));  // retire the Q loads here (hipcc otherwise drains vmcnt in front of every tile)
  __syncthreads();

  // raw scores of the tile: s[kb][i] = S[q = r][key = kv0 + 32kb + (i & 3) + 8(i >> 2) + 4h], masked where the key is not visible
  auto scores = [&](auto bufc, f32x16 (&s)[2], const int kv0) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    i32x8 kf[2][KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 a = lds_read_b128(kptr0[ks] + buf * TILE + kb * 32 * RB), b = lds_read_b128(kptr1[ks] + buf * TILE + kb * 32 * RB);
        kf[kb][ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      }
    f32x16 zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)  // (k-step outermost: the two tuples' chains alternate, no MFMA waits for its predecessor)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[kb] = mfma_f8(kf[kb][ks], qf[ks], ks == 0 ? zero : s[kb]);
    if ((CAUSAL && (kv0 + BN - 1 > qw0 + coff)) || (kv0 + BN > p.Nk)) {  // tiles that cross the diagonal or the end of the sequence
      int h4 = 4 * h;
      asm volatile("" : "+v"(h4));  // pins the limit and the compares inside this branch
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        int lim = p.Nk - 1 - kv0 - 32 * kb - h4;
        if (CAUSAL) lim = min(lim, qrow + coff - kv0 - 32 * kb - h4);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = ((i & 3) + 8 * (i >> 2) > lim) ? -INFINITY : s[kb][i];
      }
    }
  };

  auto tile = [&](auto bufc, auto firstc, const int t) {
    constexpr int buf = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    const int kv0 = t * BN;
    if (t + 1 < nT) stage_dma(t + 1, buf ^ 1);  // the next tile, in flight under this tile's arithmetic

    int always = 1;
    if constexpr (!CAUSAL) asm volatile("" : "+s"(always));  // (keeps the tile body a branch target: see fa_mfma16_kernel.hip)
    const bool wave_active = CAUSAL ? (kv0 <= qw0 + WM - 1 + coff) : (always != 0);
    if (wave_active) {
      i32x8 pb;  // P^T as the B operand of the PV product: byte j of lane half h = element j = 16kb + i of the score tuples
      float ls;
      i32x8 vf[DB];  // V^T fragments of the PV product (ONES: read under the row-sum MFMA, before its verdict)
      auto read_vt = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
              const i32x2 w = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                  (__attribute__((address_space(3))) i32x2 *)(vptr[db] + buf * TILE + (32 * kb + 8 * g) * RB));
              vf[db][4 * kb + g] = w[0];
              vf[db][4 * kb + g + 1] = w[1];
            }
      };
      bool redo = FIRST;  // wave-uniform: renew the reference first (first tile: it is -inf)
      for (;;) {
        f32x16 s[2];
        __builtin_amdgcn_s_setprio(1);
        scores(bufc, s, kv0);
        __builtin_amdgcn_s_setprio(0);
        if (__builtin_expect(redo, 0)) {
          float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
          for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);
          float lo, hi;
          half_pair(mx, lo, hi);
          mx = fmaxf(lo, hi);
          const float m_new = fmaxf(mref, mx * c2 - FA8_SHIFT);  // finite: every row sees key 0 of the first tile
          const float alpha = __builtin_amdgcn_exp2f(mref - m_new);
          l *= alpha;
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
          mref = m_new;
        }
        // P' = exp2(c.s - reference), its sum, and the packed e4m3 operand
        float ls0 = 0.0f, ls1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[0][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][i], c2, -mref));
          s[1][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[1][i], c2, -mref));
          if constexpr (!ONES) {
            ls0 = (i == 0) ? s[0][i] : ls0 + s[0][i];
            ls1 = (i == 0) ? s[1][i] : ls1 + s[1][i];
          }
        }
        ls = ls0 + ls1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // (the first conversion keeps the other half of its destination: start from an UNDEFINED register, not from a zeroed one --
            //  the `= 0` cost a v_mov per packed dword, 8 of the hot tile's 90 vector instructions)
            int w;
            asm volatile("" : "=v"(w));
            w = __builtin_amdgcn_cvt_pk_fp8_f32(s[kb][4 * g + 0], s[kb][4 * g + 1], w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(s[kb][4 * g + 2], s[kb][4 * g + 3], w, true);
            pb[4 * kb + g] = w;
          }
        bool stale;  // wave-uniform
        if constexpr (ONES) {
          // the tile's row sums, ones . P^T: every register of the tuple = the sum of row r's 64 ROUNDED probabilities. A probability
          // above e4m3's 448 has converted to NaN (v_cvt_pk_fp8_f32 does not saturate) and shows here as a NaN sum: the stale-reference test
          f32x16 zero;
#pragma unroll
          for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
          __builtin_amdgcn_s_setprio(1);
          const f32x16 ts = mfma_f8(ones8, pb, zero);
          read_vt();
          __builtin_amdgcn_s_setprio(0);
          ls = ts[0];
          // (an integer test of the bit pattern: the file is compiled with -fno-honor-nans, under which `ls != ls` folds to false)
          stale = __builtin_amdgcn_ballot_w64((__builtin_bit_cast(unsigned, ls) & 0x7fffffffu) > 0x7f800000u) != 0;
        } else {
          stale = __builtin_amdgcn_ballot_w64(!(ls <= FA8_SUM_LIMIT)) != 0;  // (a NaN sum counts as stale)
        }
        if (__builtin_expect(!stale || redo, 1)) break;  // (after a renewal every P' <= 8: a second stale reading is inf / NaN input)
        redo = true;
      }
      l += ls;
      // ---- O^T += V^T.P^T: one MFMA per 32-wide d block, 64 keys deep
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!ONES) read_vt();
#pragma unroll
      for (int db = 0; db < DB; ++db) oacc[db] = mfma_f8(vf[db], pb, oacc[db]);
      __builtin_amdgcn_s_setprio(0);
    }
    if (t + 1 < nT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces issued at the top of this tile have landed
    __syncthreads();
  };
  tile(std::integral_constant<int, 0>{}, std::true_type{}, 0);
  for (int t = 1; t < nT; t += 2) {
    tile(std::integral_constant<int, 1>{}, std::false_type{}, t);
    if (t + 1 < nT) tile(std::integral_constant<int, 0>{}, std::false_type{}, t + 1);
  }

  // ---- epilogue: normalise, LSE, O tile (bf16) -> LDS -> coalesced 16-byte stores
  lds_char *Ot = smem + wave * (WM * ORB);  // this wave's [32][D] bf16 tile (the K / V buffers are free since the last barrier)
  if constexpr (!ONES) {
    float lo, hi;
    half_pair(l, lo, hi);
    l = lo + hi;
  }
  const float inv_l = __builtin_amdgcn_rcpf(l);
  if (p.lse != nullptr && h == 0 && qrow < p.N) p.lse[(long long)bh * p.N + qrow] = (mref + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      // registers 4g4 .. 4g4+3 = d columns 32db + 8g4 + 4h + 0..3 of row r
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 e;
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (__bf16)(oacc[db][4 * g4 + i] * inv_l);
      const int col_b = (32 * db + 8 * g4 + 4 * h) * 2;
      const int ch = (col_b >> 4) ^ (r & (CPO - 1));
      lds_write_b64(Ot + r * ORB + (ch << 4) + (col_b & 15), __builtin_bit_cast(u32x2, e));
    }
  __syncthreads();
  {
    __bf16 *Og = (__bf16 *)p.o + base;
#pragma unroll
    for (int it = 0; it < WM * CPO / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPO, ch = idx % CPO;
      const u32x4 vv = lds_read_b128(Ot + row * ORB + ((ch ^ (row & (CPO - 1))) << 4));
      if (qw0 + row < p.N) *reinterpret_cast<u32x4 *>(Og + (long long)(qw0 + row) * D + ch * 8) = vv;
    }
  }
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(NTHREADS, (D == 64 ? FA8_OCC : FA8_OCC128)) void fwd_fp8_kernel(Params p) {
  fwd_fp8_body<D, CAUSAL>(p);
}

bool fp8pv_supported(int dtype, int D) { return dtype == FA_DTYPE_FP8_E4M3 && (D == 64 || D == 128); }

template <int D, bool CAUSAL>
static hipError_t launch8_one(const Params &p, hipStream_t s) {
  const int nQ = (p.N + BM - 1) / BM;
  const size_t smem = std::max((size_t)4 * BN * D, (size_t)BM * 2 * D);  // K / V double buffers; the epilogue's bf16 O tiles
  auto kern = fwd_fp8_kernel<D, CAUSAL>;
  Params pp = p;
  pp.head_group = causal_head_group(p, D, 1);
  set_block_divisors(pp, nQ, pp.head_group);
  (void)hipGetLastError();  // do not report an older sticky error as this launch's
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, pp);
  return hipGetLastError();
}

hipError_t launch_fp8pv(const Params &p, int dtype, hipStream_t s) {
  if (!fp8pv_supported(dtype, p.D)) return hipErrorInvalidValue;
  if (p.D == 128) return p.is_causal ? launch8_one<128, true>(p, s) : launch8_one<128, false>(p, s);
  return p.is_causal ? launch8_one<64, true>(p, s) : launch8_one<64, false>(p, s);
}

}  // namespace fa
