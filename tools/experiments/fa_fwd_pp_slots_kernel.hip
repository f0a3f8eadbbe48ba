// fa_fwd_pp_kernel.hip -- the operator on the CDNA4 matrix cores, paired-block ("ping-pong") form.
//
// Same math and the same operand maps as fa_mfma_kernel.hip (replaces
// /root/reference/kernels.metal:600-883): tiled QK^T -> online softmax -> PV, causal predicate
// `key > query -> masked` (kernels.metal:748), whole-tile skip (kernels.metal:682),
// L = m + ln(l) (kernels.metal:862-864); fp32 accumulators for S, O, m, l.
//
// What is different is WHO overlaps with WHOM. In fa_mfma_kernel.hip a wave runs
// QK^T -> softmax -> PV strictly one after another and relies on other waves of the SIMD to fill
// the matrix pipe during its softmax; measured MFMA/VALU co-execution there is 33 %
// (profiles/r01_final/pmc_summary.json). Here one wave owns TWO independent 32-row query blocks
// (A, B) and its single instruction stream alternates
//      slot 1:  MFMA  QK^T(A,t+1), PV(A,t)    ||   VALU  softmax(B,t)
//      slot 2:  MFMA  QK^T(B,t+1), PV(B,t)    ||   VALU  softmax(A,t+1)
// with the softmax cut into pieces that are placed, in program order, in the issue gaps between
// the MFMAs (an MFMA occupies the matrix pipe for 32 cycles but the issue port for 8;
// MI355X_MICROARCH.md 'vector-instruction ISSUE cost'). Nothing of block X is touched by the
// matrix pipe while X's softmax runs, so the overlap is by construction, not by luck of wave
// arbitration. sched_barrier(0) between the groups and empty volatile asm "pins" on the softmax
// results keep the order the source states (LLVM otherwise sinks the exp/pack work to its first use,
// one slot later, out of the gaps).
//
//   * workgroup = 4 waves = 256 query rows of one (batch, head); wave = 64 rows = blocks A, B;
//     one wave per SIMD with the whole register file: O^T lives in asm-owned accumulation
//     registers (fa_mfma_common.h), S / P / Q in architectural VGPRs
//   * K/V tiles of 64 keys double-buffered in LDS; one iteration consumes V(t) and K(t+1), so the
//     loads of K(t+2), V(t+1) are issued at its start and written to the free buffers at its end:
//     ONE barrier per 2 slots (32 MFMAs per wave at head_dim 64, 64 at head_dim 128)
//   * hot iterations (tiles that need no mask for either block, not the wave's last) carry no mask
//     code and a compile-time LDS buffer index; the one or two remaining iterations of a wave run
//     a cold variant that always applies the mask
//   * deferred row max (T13): O/l are rescaled only when some row's tile max exceeds the running
//     reference by more than 2^THR; p = exp2(c.s - c.m_ref) <= 2^THR otherwise. m_ref, l and O
//     stay mutually consistent, LSE = m_ref.scale + ln(l) is exact either way.
#include "fa_mfma_common.h"

#ifndef FA_PP_THR
#define FA_PP_THR 8.0f  // log2 units: P values are bounded by 2^8 between rescales
#endif
#ifndef FA_PP_LA
#define FA_PP_LA 2      // LDS operand reads are issued this many MFMAs ahead of their use
#endif

namespace fa {

constexpr int PP_BM = 256;  // query rows per workgroup
constexpr int PP_WM = 64;   // query rows per wave (two 32-row blocks)

template <typename Tag, int D, bool CAUSAL>
__global__ __launch_bounds__(NTHREADS, 1) void fwd_pp_kernel(Params p) {
  using M = MT<Tag>;
  using elem = typename M::elem;
  typedef elem elem2 __attribute__((ext_vector_type(2)));
  constexpr int RB = D * 2;                 // LDS row bytes
  constexpr int CPR = D / 8;                // 16-byte chunks per row
  constexpr int KS = D / 16;                // k-steps of the QK^T product
  constexpr int DB = D / 32;                // 32-wide d blocks of O^T
  constexpr int NACC = 2 * DB * 16;         // accumulation registers owned: O^T of block x, d block db = a[16(x DB + db) ..+15]
  constexpr int TILE = BN * RB;             // bytes of one K (or V) tile in LDS
  constexpr bool IS_FP8 = std::is_same<Tag, FP8>::value;
  constexpr int GB = IS_FP8 ? 1 : 2;
  constexpr int GRB = D * GB;               // global row bytes
  constexpr int GTILE = BN * GRB;
  constexpr int GCPR = GRB / 16;
  constexpr int NCH = BN * GCPR / NTHREADS; // staged 16-byte global chunks per thread per tile
  constexpr int LA = FA_PP_LA;
  static_assert(NACC == 64 || NACC == 128, "head_dim 64 or 128");

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  lds_char *Kbuf = smem;             // [2][BN][RB], rows swizzled
  lds_char *Vbuf = smem + 2 * TILE;  // [2][BN][RB], rows swizzled

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  const int nQ = (p.N + PP_BM - 1) / PP_BM;
  const int BH = p.B * p.H;
  int bh, qb;
  map_block<CAUSAL>(blockIdx.x, BH, nQ, bh, qb, p.head_group);
  const long long base = (long long)(bh / p.H) * p.batch_stride + (long long)(bh % p.H) * p.head_stride;
  const long long base_kv = (long long)(bh / p.H) * p.kv_batch_stride + (long long)((bh % p.H) / (p.H / p.Hkv)) * p.kv_head_stride;
  const int coff = p.Nk - p.N;  // bottom-right aligned causal mask for Nq != Nk
  const int q0 = qb * PP_BM;
  const int qw0 = q0 + wave * PP_WM;

  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.q + base * GB), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.k + base_kv * GB), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.v + base_kv * GB), 0, kv_head_bytes, 0x00020000);

  // ---- Q fragments of both blocks (B operand of K.Q^T): lane (r,h) holds Q[row][16ks+8h .. +7]
  u32x4 qf[2][KS];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned row = (unsigned)(qw0 + 32 * x + r);
      if constexpr (IS_FP8) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rq, row * GRB + (2 * ks + h) * 8, 0, 0);
        qf[x][ks] = fp8x8_to_bf16(t);
      } else {
        qf[x][ks] = __builtin_amdgcn_raw_buffer_load_b128(rq, row * RB + (2 * ks + h) * 16, 0, 0);
      }
    }

  // ---- per-lane LDS offsets (same images as fa_mfma_kernel.hip)
  const int kx = (D == 64) ? ((r >> 1) & 7) : (r & 15);
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = r * RB + (((2 * ks + h) ^ kx) << 4);
  const int g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  const int vx = (D == 64) ? (((vq >> 1) & 1) << 2) : (vq << 2);
  int voff[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
    voff[db] = (4 * h + vq) * RB + ((((4 * db) ^ vx) + 2 * g1 + (vp >> 1)) << 4) + 8 * (vp & 1);

  // ---- staging map: thread -> NCH 16-byte global chunks of a tile
  int st_g[NCH], st_k[NCH], st_v[NCH], st_k1[IS_FP8 ? NCH : 1], st_v1[IS_FP8 ? NCH : 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHREADS;
    const int row = c / GCPR, gch = c % GCPR;
    st_g[i] = row * GRB + gch * 16;
    const int skx = (D == 64) ? ((row >> 1) & 7) : (row & 15);
    const int svx = (D == 64) ? (((row >> 1) & 1) << 2) : ((row & 3) << 2);
    const int ch = IS_FP8 ? 2 * gch : gch;
    st_k[i] = row * RB + ((ch ^ skx) << 4);
    st_v[i] = row * RB + ((ch ^ svx) << 4);
    if constexpr (IS_FP8) {
      st_k1[i] = row * RB + (((ch + 1) ^ skx) << 4);
      st_v1[i] = row * RB + (((ch + 1) ^ svx) << 4);
    }
  }

  // tiles: the workgroup stages nT tiles; this wave computes the first nTw of them, and its first
  // nClean iterations are "hot" (neither tile t nor t+1 needs a mask for either block, t+1 < nTw)
  const int kv_end = CAUSAL ? min(p.Nk, q0 + PP_BM + coff) : p.Nk;
  const int nT = (kv_end + BN - 1) / BN;
  const int kv_end_w = CAUSAL ? min(p.Nk, qw0 + PP_WM + coff) : p.Nk;
  const int nTw = (kv_end_w + BN - 1) / BN;  // >= 1 (causal needs Nk >= Nq, so key 0 is visible to every row)
  int n_unmasked = p.Nk / BN;                // tile u is unmasked iff 64u+64 <= Nk and (causal) 64u+63 <= qw0+coff
  if (CAUSAL) n_unmasked = min(n_unmasked, (qw0 + coff >= BN - 1) ? (qw0 + coff - (BN - 1)) / BN + 1 : 0);
  const int nClean = max(0, n_unmasked - 1);

  u32x4 kst[NCH], vst[NCH];
  auto load_k = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)t * GTILE + st_g[i], 0, 0);
  };
  auto load_v = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)t * GTILE + st_g[i], 0, 0);
  };
  auto write_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if constexpr (IS_FP8) {
        lds_write_b128(Kbuf + buf * TILE + st_k[i], fp8x8_to_bf16(u32x2{kst[i][0], kst[i][1]}));
        lds_write_b128(Kbuf + buf * TILE + st_k1[i], fp8x8_to_bf16(u32x2{kst[i][2], kst[i][3]}));
      } else {
        lds_write_b128(Kbuf + buf * TILE + st_k[i], kst[i]);
      }
    }
  };
  auto write_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if constexpr (IS_FP8) {
        lds_write_b128(Vbuf + buf * TILE + st_v[i], fp8x8_to_bf16(u32x2{vst[i][0], vst[i][1]}));
        lds_write_b128(Vbuf + buf * TILE + st_v1[i], fp8x8_to_bf16(u32x2{vst[i][2], vst[i][3]}));
      } else {
        lds_write_b128(Vbuf + buf * TILE + st_v[i], vst[i]);
      }
    }
  };

  // ---- per-block state (index 0 = A, 1 = B; every use below has a compile-time index)
  f32x16 s[2][2];      // score tiles [block][kb]
  u32x4 pfr[2][2][2];  // P fragments [block][kb][st]: 8 x 16-bit = the B operand of one PV step
  float mref[2], mthr[2], negmc[2], l0[2], l1[2];
  const float c2 = p.scale * 1.4426950408889634f;  // scale * log2(e)
  const float thr_raw = FA_PP_THR / c2;            // the threshold in raw-score units
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    mref[x] = -INFINITY;
    mthr[x] = -INFINITY;
    negmc[x] = 0.0f;  // replaced by the first tile's rescale before any use
    l0[x] = 0.0f;
    l1[x] = 0.0f;
  }
  static_for<0, NACC>([&](auto ic) __attribute__((always_inline)) { acc_zero1<NACC, decltype(ic)::value>(); });

  // ---- prologue: K(0), V(0), K(1) into LDS
  {
    // all three tiles in flight together (registers are plentiful before the loop); a tile past the
    // end of the head reads as zero through the descriptor's range check and is never used
    u32x4 k1[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) k1[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)GTILE + st_g[i], 0, 0);
    load_k(0);
    load_v(0);
    write_k(0);
    write_v(0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) kst[i] = k1[i];
    write_k(1);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[x][ks]));  // retire the Q loads here (see fa_mfma_kernel.hip)
  __syncthreads();

  // One slot: [QK^T(Y, tile in Kbuf[kbuf])] [PV(Y, tile in Vbuf[vbuf])] on the matrix pipe, softmax of
  // block X (scores of tile tX already in s[X]) in the issue gaps. Flags select the parts that exist;
  // vbufc / kbufc are integral_constants (hot path: LDS addresses are base register + immediate) or ints.
  // Each gap group is  { VALU piece ; MFMA ; LDS reads for the MFMA LA steps later }.
  auto slot = [&](auto vbufc, auto kbufc, auto xc, auto haspvc, auto hasqkc, auto hassmc, auto maskc, const int tX) __attribute__((always_inline)) {
    constexpr int X = decltype(xc)::value, Y = 1 - X;
    constexpr bool HAS_PV = decltype(haspvc)::value, HAS_QK = decltype(hasqkc)::value, HAS_SM = decltype(hassmc)::value;
    constexpr bool MASK = decltype(maskc)::value;
    constexpr int NQK = HAS_QK ? 2 * KS : 0, NPV = HAS_PV ? 4 * DB : 0, NM = NQK + NPV;
    constexpr int GM = HAS_SM ? (NM >= 32 ? 4 : 2) : 0;  // groups that carry the row-max pieces
    constexpr int NU = 16;                               // exp/sum/pack units of two scores each
    const int vbuf = vbufc, kbuf = kbufc;
    const lds_char *Vt = Vbuf + vbuf * TILE;
    const lds_char *Kt = Kbuf + kbuf * TILE;

    // ---- mask S_X (cold variants only): masked iff key > query (kernels.metal:748) or key >= Nk
    if constexpr (HAS_SM && MASK) {
      const int kv0 = tX * BN;
      const int qx0 = qw0 + 32 * X;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        int lim = p.Nk - 1 - kv0 - 32 * kb - 4 * h;
        if (CAUSAL) lim = min(lim, qx0 + r + coff - kv0 - 32 * kb - 4 * h);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kpart = (i & 3) + 8 * (i >> 2);
          s[X][kb][i] = (kpart > lim) ? -INFINITY : s[X][kb][i];
        }
      }
    }

    // ---- matrix-pipe work list (QK^T first: its results then sit >= 4*DB MFMAs away from their first reader)
    u32x4 aop[NM > 0 ? NM : 1];
    auto opread = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
#ifdef FA_PP_DIAG
      if constexpr ((FA_PP_DIAG & 8) != 0) {  // no LDS reads: operands are whatever the registers hold
        asm volatile("" : "=v"(aop[g]));
        return;
      }
#endif
      if constexpr (g < NQK) {
        aop[g] = lds_read_b128(Kt + (g / KS) * 32 * RB + koff[g % KS]);
      } else {
        constexpr int j = g - NQK;
        constexpr int kb = j / (2 * DB), st = (j / DB) % 2, db = j % DB;
        const lds_char *vb = Vt + (32 * kb + 16 * st) * RB + voff[db];
        const s16x4 lo = lds_read_tr16(vb), hi = lds_read_tr16(vb + 8 * RB);
        aop[g] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto domfma = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
#ifdef FA_PP_DIAG
      if constexpr ((FA_PP_DIAG & 32) != 0) {  // no MFMAs
        asm volatile("" ::"v"(aop[g]));
        if constexpr (g < NQK && (g % KS) == 0) asm volatile("" : "=v"(s[Y][g / KS]));
        return;
      }
#endif
      if constexpr (g < NQK) {
        if constexpr ((g % KS) == 0) M::mfma_v0(s[Y][g / KS], aop[g], qf[Y][g % KS]);
        else M::mfma_v(s[Y][g / KS], aop[g], qf[Y][g % KS]);
      } else {
        constexpr int j = g - NQK;
        constexpr int kb = j / (2 * DB), st = (j / DB) % 2, db = j % DB;
        acc_mfma<Tag, NACC, Y * DB + db>(aop[g], pfr[Y][kb][st]);
      }
    };

    // ---- softmax pieces of block X. One wave per SIMD: nothing but this wave's own independent
    // instructions hides VALU / transcendental latency, so the work is cut into stages that sit a whole
    // MFMA gap apart: 4 interleaved max chains; fma | exp2 | sum+pack of consecutive units in different gaps.
    float mxc[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // chain c covers s[X][c/2][8(c%2) .. +7]
    auto max_op = [&](int k) __attribute__((always_inline)) {  // 16 ops: k%4 = chain, k/4 = step
      const int c = k % 4, stp = k / 4, kb = c / 2, e0 = 8 * (c % 2);
      const f32x16 &t = s[X][kb];
      if (stp == 0) mxc[c] = fmaxf(fmaxf(t[e0], t[e0 + 1]), t[e0 + 2]);
      else if (stp < 3) mxc[c] = fmaxf(fmaxf(mxc[c], t[e0 + 2 * stp + 1]), t[e0 + 2 * stp + 2]);  // -> v_max3_f32
      else mxc[c] = fmaxf(mxc[c], t[e0 + 7]);
    };
    auto unit_fma = [&](int u) __attribute__((always_inline)) {
      const int kb = u / 8, e = 2 * (u % 8);
      s[X][kb][e] = __builtin_fmaf(s[X][kb][e], c2, negmc[X]);
      s[X][kb][e + 1] = __builtin_fmaf(s[X][kb][e + 1], c2, negmc[X]);
    };
    auto unit_exp = [&](int u) __attribute__((always_inline)) {
      const int kb = u / 8, e = 2 * (u % 8);
      s[X][kb][e] = __builtin_amdgcn_exp2f(s[X][kb][e]);
      s[X][kb][e + 1] = __builtin_amdgcn_exp2f(s[X][kb][e + 1]);
    };
    auto unit_sum = [&](int u) __attribute__((always_inline)) {  // row-sum partials, one packed P dword
      const int kb = u / 8, e = 2 * (u % 8);
      l0[X] += s[X][kb][e];
      l1[X] += s[X][kb][e + 1];
      elem2 pk;
      pk[0] = (elem)s[X][kb][e];
      pk[1] = (elem)s[X][kb][e + 1];
      const unsigned w = __builtin_bit_cast(unsigned, pk);
      // pin (inputs only, so hipcc pads nothing behind it): LLVM otherwise sinks the work to the
      // first use of P (the next slot), out of this gap
      asm volatile("" ::"v"(w), "v"(l0[X]), "v"(l1[X]));
      pfr[X][kb][e / 8][(e % 8) / 2] = w;
    };
    // row r of the pipelined order: sum(r-2), exp(r-1), fma(r)   (r = 0 .. NU+1)
    auto unit_row = [&](int rr) __attribute__((always_inline)) {
      if (rr >= 2 && rr - 2 < NU) unit_sum(rr - 2);
      if (rr >= 1 && rr - 1 < NU) unit_exp(rr - 1);
      if (rr < NU) unit_fma(rr);
    };
    constexpr int NR = NU + 2;

    if constexpr (NM > 0) {
      static_for<0, (LA < NM ? LA : NM)>([&](auto gc) __attribute__((always_inline)) { opread(gc); });
      __builtin_amdgcn_sched_barrier(0);
    }
    // segment 1: the first GM groups carry the row maximum
    static_for<0, GM>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
#pragma unroll
      for (int k = g * 16 / GM; k < (g + 1) * 16 / GM; ++k) max_op(k);
      if constexpr (g < NM) {
        __builtin_amdgcn_sched_barrier(0);
        domfma(gc);
        if constexpr (g + LA < NM) opread(std::integral_constant<int, g + LA>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (HAS_SM) {
      float mx = fmaxf(fmaxf(mxc[0], mxc[1]), fmaxf(mxc[2], mxc[3]));
      {
        float lo, hi;
        half_pair(mx, lo, hi);
        mx = fmaxf(lo, hi);
      }
      // deferred max: rescale only when some row of the block outgrew its reference by > 2^THR.
      // O^T of block X is in the accumulation file: the matrix pipe wrote it last one whole slot ago.
      if (__builtin_amdgcn_ballot_w64(mx > mthr[X]) != 0) {
        const float m_new = fmaxf(mref[X], mx);
        const float alpha = __builtin_amdgcn_exp2f((mref[X] - m_new) * c2);  // first tile: exp2(-inf) = 0
        l0[X] *= alpha;
        l1[X] *= alpha;
        asm volatile("s_nop 15\n\ts_nop 7" ::"v"(alpha));  // MFMA write -> accvgpr read, VALU write -> asm read
        static_for<0, 16 * DB>([&](auto ic) __attribute__((always_inline)) { acc_scale1<NACC, X * DB * 16 + decltype(ic)::value>(alpha); });
        asm volatile("s_nop 3");                           // accvgpr write -> MFMA read as C
        mref[X] = m_new;
        mthr[X] = m_new + thr_raw;
        negmc[X] = -m_new * c2;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // segment 2: the remaining groups carry the exp / sum / pack units, spread evenly
    constexpr int NG2 = NM - GM;  // groups left (may be <= 0)
    if constexpr (NG2 > 0) {
      static_for<GM, NM>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (HAS_SM) {
#pragma unroll
          for (int rr = (g - GM) * NR / NG2; rr < (g - GM + 1) * NR / NG2; ++rr) unit_row(rr);
          __builtin_amdgcn_sched_barrier(0);
        }
        domfma(gc);
        if constexpr (g + LA < NM) opread(std::integral_constant<int, g + LA>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    } else if constexpr (HAS_SM) {
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) unit_row(rr);
    }
    // a slot that ends on QK^T: keep the asm MFMA's result 16 wait states away from its first reader
    if constexpr (HAS_QK && !HAS_PV) asm volatile("s_nop 15");
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using T = std::true_type;
  using F = std::false_type;

  // ---- pipeline fill: S_A(0), then S_B(0) under softmax(A,0)   (K(0) is in Kbuf[0])
  slot(I0{}, I0{}, I1{}, F{}, T{}, F{}, F{}, 0);  // Y = A: QK^T(A,0) only
  slot(I0{}, I0{}, I0{}, F{}, T{}, T{}, T{}, 0);  // Y = B: QK^T(B,0) || softmax(A,0), masked variant

  // One iteration: consumes V(t) = Vbuf[PAR] and K(t+1) = Kbuf[PAR^1]; stages K(t+2) -> Kbuf[PAR], V(t+1) -> Vbuf[PAR^1]
  auto iter = [&](auto parc, const int t) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    using PV = std::integral_constant<int, PAR>;
    using PK = std::integral_constant<int, PAR ^ 1>;
#ifdef FA_PP_DIAG  // timing experiments only (results are wrong): 1 = no barrier, 2 = no global loads, 4 = no LDS writes
    const bool ldk = !(FA_PP_DIAG & 2) && t + 2 < nT, ldv = !(FA_PP_DIAG & 2) && t + 1 < nT;
#else
    const bool ldk = t + 2 < nT, ldv = t + 1 < nT;
#endif
    if (ldk) load_k(t + 2);
    if (ldv) load_v(t + 1);
    if (t < nClean) {                                       // hot: no masks, static LDS buffers
#if defined(FA_PP_DIAG) && (FA_PP_DIAG & 16)               // timing experiment: no softmax
      slot(PV{}, PK{}, I1{}, T{}, T{}, F{}, F{}, t);
      slot(PV{}, PK{}, I0{}, T{}, T{}, F{}, F{}, t + 1);
#else
      slot(PV{}, PK{}, I1{}, T{}, T{}, T{}, F{}, t);        // QK^T(A,t+1), PV(A,t) || softmax(B,t)
      slot(PV{}, PK{}, I0{}, T{}, T{}, T{}, F{}, t + 1);    // QK^T(B,t+1), PV(B,t) || softmax(A,t+1)
#endif
    } else if (t < nTw) {                                   // cold: masks always applied, runtime LDS buffers
      const int vb = PAR, kb = PAR ^ 1;
      if (t + 1 < nTw) {
        slot(vb, kb, I1{}, T{}, T{}, T{}, T{}, t);
        slot(vb, kb, I0{}, T{}, T{}, T{}, T{}, t + 1);
      } else {                                              // this wave's last tile: drain
        slot(vb, kb, I1{}, T{}, F{}, T{}, T{}, t);          // PV(A,t) || softmax(B,t)
        slot(vb, kb, I0{}, T{}, F{}, F{}, F{}, t);          // PV(B,t)
      }
    }
#ifdef FA_PP_DIAG
    if (!(FA_PP_DIAG & 4)) {
      if (ldk || (FA_PP_DIAG & 2)) write_k(PAR);
      if (ldv || (FA_PP_DIAG & 2)) write_v(PAR ^ 1);
    }
    if (!(FA_PP_DIAG & 1)) __syncthreads();
#else
    if (ldk) write_k(PAR);
    if (ldv) write_v(PAR ^ 1);
    __syncthreads();
#endif
  };
  for (int t = 0; t < nT; t += 2) {
    iter(I0{}, t);
    if (t + 1 < nT) iter(I1{}, t + 1);
  }

  // ---- epilogue: normalise, LSE, O tiles -> LDS -> whole rows, 16 B per lane
  asm volatile("s_nop 15\n\ts_nop 7");        // last PV MFMA -> accvgpr reads
  lds_char *Ot = smem + wave * (PP_WM * RB);  // this wave's [64][D] tile (inside the K/V buffers; all reads are done)
  elem *Og = (elem *)p.o + base;
  static_for<0, 2>([&](auto xc) __attribute__((always_inline)) {
    constexpr int x = decltype(xc)::value;
    float l = l0[x] + l1[x];
    {
      float lo, hi;
      half_pair(l, lo, hi);
      l = lo + hi;
    }
    const float inv_l = 1.0f / l;
    const int qrow = qw0 + 32 * x + r;
    if (p.lse != nullptr && h == 0 && qrow < p.N) p.lse[(long long)bh * p.N + qrow] = mref[x] * p.scale + logf(l);
    static_for<0, DB * 4>([&](auto jc) __attribute__((always_inline)) {
      constexpr int db = decltype(jc)::value / 4, g4 = decltype(jc)::value % 4;
      constexpr int R0 = 16 * (x * DB + db) + 4 * g4;
      elem2 a, b;
      a[0] = (elem)(acc_read1<NACC, R0 + 0>() * inv_l);
      a[1] = (elem)(acc_read1<NACC, R0 + 1>() * inv_l);
      b[0] = (elem)(acc_read1<NACC, R0 + 2>() * inv_l);
      b[1] = (elem)(acc_read1<NACC, R0 + 3>() * inv_l);
      u32x2 w;
      w[0] = __builtin_bit_cast(unsigned, a);
      w[1] = __builtin_bit_cast(unsigned, b);
      const int col_b = (32 * db + 8 * g4 + 4 * h) * 2;
      const int ch = (col_b >> 4) ^ (r & (CPR - 1));
      lds_write_b64(Ot + (32 * x + r) * RB + (ch << 4) + (col_b & 15), w);
    });
  });
  __syncthreads();
#pragma unroll
  for (int it = 0; it < PP_WM * CPR / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / CPR, ch = idx % CPR;
    const u32x4 vv = lds_read_b128(Ot + row * RB + ((ch ^ (row & (CPR - 1))) << 4));
    if (qw0 + row < p.N) *reinterpret_cast<u32x4 *>(Og + (long long)(qw0 + row) * D + ch * 8) = vv;
  }
}

// ---------------------------------------------------------------------------
bool pp_supported(int dtype, int D) {
  return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16 || dtype == FA_DTYPE_FP8_E4M3) && (D == 64 || D == 128);
}

template <typename Tag, int D, bool CAUSAL>
static hipError_t launch_pp_one(const Params &p, hipStream_t s) {
  const int nQ = (p.N + PP_BM - 1) / PP_BM;
  const size_t smem = 4 * BN * D * 2;
  auto kern = fwd_pp_kernel<Tag, D, CAUSAL>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem);
    if (e != hipSuccess) return e;
  }
  Params pp = p;
  pp.head_group = causal_head_group(p, D, std::is_same<Tag, FP8>::value ? 1 : 2);
  (void)hipGetLastError();  // do not report an older sticky error as this launch's
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, pp);
  return hipGetLastError();
}

template <typename Tag>
static hipError_t launch_pp_dt(const Params &p, hipStream_t s) {
  if (p.D == 64) return p.is_causal ? launch_pp_one<Tag, 64, true>(p, s) : launch_pp_one<Tag, 64, false>(p, s);
  return p.is_causal ? launch_pp_one<Tag, 128, true>(p, s) : launch_pp_one<Tag, 128, false>(p, s);
}

hipError_t launch_pp(const Params &p, int dtype, hipStream_t s) {
  if (dtype == FA_DTYPE_FP8_E4M3) return launch_pp_dt<FP8>(p, s);
  return dtype == FA_DTYPE_F16 ? launch_pp_dt<F16>(p, s) : launch_pp_dt<BF16>(p, s);
}

}  // namespace fa
