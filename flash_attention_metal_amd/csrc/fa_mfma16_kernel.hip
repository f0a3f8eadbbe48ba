// fa_mfma16_kernel.hip -- the operator on v_mfma_f32_16x16x32_{bf16,f16} (variant "mfma16").
//
// Same math and the same workgroup shape as fa_mfma_kernel.hip (replaces /root/reference/kernels.metal:600-883: tiled
// QK^T -> online softmax -> PV, causal predicate `key > query -> masked` kernels.metal:748, L = m + ln(l)
// kernels.metal:862-864), but every matrix product runs on the 16x16x32 instruction instead of 32x32x16. Why a second
// kernel for the same work: on random data the chip is power-limited under these loops and holds a higher clock on
// the 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x the FLOP/s of the 32x32x16 loop at equal
// cycles per FLOP; cdna_hip_programming.md rule 28) -- so cycles per FLOP do not decide which is faster, wall does.
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); a wave owns 32 rows = TWO 16-row query tiles,
// so every K / V^T fragment read from LDS still feeds two MFMAs -- LDS traffic per FLOP equals the 32x32x16 kernel's):
//   * S^T = K.Q^T per 16 keys x 16 queries ("swapped" product): lane (c = lane & 15, g = lane >> 4) holds the scores of
//     query c of each query tile against keys 16kt + 4g + i (i = register) -- a lane serves TWO query rows, 16 scores each
//     per 64-key tile; row max / row sum are in-register reductions, the four lanes of a row (g = 0..3) meet only on the
//     rare exact path and in the epilogue
//   * pre-scaled query operand Q~ = round(scale.log2e.Q) and the running reference -m as the C operand of the first MFMA
//     of every score chain (4 registers per query tile instead of 16): P = exp2(S') with no FMA; the row sums decide
//     whether m is stale (sum-triggered deferred max, threshold 2^8) exactly as in fa_mfma_kernel.hip
//   * P^T feeds PV straight from the score registers: the B operand's k index (8g + j) is key 16(j >> 2) + 4g + (j & 3) of a
//     32-key step, and V^T is read with ds_read_b64_tr_b16 in the same order (V stays row-major in HBM and LDS)
//   * K/V tiles of 64 keys double-buffered in LDS, global -> LDS by LDS-DMA with the chunk swizzle on the source address;
//     one barrier per tile; images: K chunk ^ ((row >> 1) & 7), V chunk ^ (((row >> 1) & 3) << 1) at head_dim 64 (both
//     conflict-free for these reads: checked on the CPU by tests/test_lds_images.py)
//   * epilogue: O tile -> LDS -> whole rows, 16 B per lane; LSE = m.ln2 + ln(l)
#include <stdlib.h>

#include <algorithm>

#include "fa_mfma_common.h"

#ifndef FA16_DEFER_THR
#define FA16_DEFER_THR 8.0f  // log2 units: P <= 2^8 between rescales
#endif
#ifndef FA16_LAK
#define FA16_LAK (D == 128 ? 3 : 2)  // K fragments are read this many fragments (each feeds two MFMAs) ahead of their use (head_dim 128: 3,
                                    // config 4 +1.5 % over 2, profiles/r04/ab_mfma16_d128_knobs.log)
#endif
#ifndef FA16_LAV
#define FA16_LAV (D == 128 ? 3 : 2)  // ... and V^T fragments
#endif
#ifndef FA16_PRIO
#define FA16_PRIO 1  // 1: wave priority raised around the MFMA clusters
#endif
#ifndef FA16_ONES
#define FA16_ONES 1  // 1: the row sums come out of the matrix core (a fifth "d tile" of ones in the PV product: +4 MFMAs per 64-key tile) and
                     // the staleness test reads the exponent bits of the packed probabilities (8 v_or3 instead of 32 v_add): the loop is
                     // VALU-issue-bound with the matrix pipe half idle, so VALU is traded for MFMA; 0: row sums by v_add, tested against 2^THR
#endif
#ifndef FA16_HALVES
#define FA16_HALVES 0  // 1: the hot pass works on one 32-key half at a time (16 live score registers instead of 32)
#endif
#ifndef FA16_BN128
#define FA16_BN128 64  // keys per staged tile of the head_dim-128 instantiation (64: 64 KiB of LDS, two workgroups per CU; 32: 32 KiB)
#endif
#ifndef FA16_OCC128
#define FA16_OCC128 2  // workgroups per CU the head_dim-128 kernel is compiled for (2: 256 registers, 3: 168)
#endif
#ifndef FA16_SUB8
#define FA16_SUB8 1  // tiles per staged unit (barrier) of the eight-wave head_dim-64 instantiation: 1, or 2 (64 KiB of LDS)
#endif
#ifndef FA16_OCC
#define FA16_OCC 4  // workgroups per CU the head_dim-64 kernel is compiled for (128 registers)
#endif

namespace fa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename Tag> struct MT16;
template <> struct MT16<BF16> {
  using elem = __bf16;
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MT16<F16> {
  using elem = _Float16;
  using vec8 = f16x8;
  __device__ static __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// value of lane ^ 16 / lane ^ 32 (rare path and epilogue only: ds_bpermute, no LDS memory)
__device__ __forceinline__ float xlane(float x, int mask) { return __shfl_xor(x, mask, 64); }

// PAD: head dims other than 64 / 128 (any multiple of 8 below 128 that has no kernel of its own: 40, 48, 72, 80, 112, ...) run this
// instantiation on ZERO-PADDED rows, as the backward does (fa_bwd_kernels.hip, FA_BWD_PAD): rows keep their packed pitch of p.D elements
// in global memory, the LDS images and register fragments have the kernel's pitch, and every 16-byte chunk at or past column p.D is
// fetched from an offset outside the buffer descriptor's range, which reads as zeros for register loads and LDS-DMA alike. The padding
// adds 0 to every score and its O columns are never stored.
template <typename Tag, int D, bool CAUSAL, int RW, bool PAD, int SUB>
__device__ __forceinline__ void fwd_mfma16_body(const Params &p) {
  using M = MT16<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  constexpr int BMR = RW * WM;     // query rows of the block (RW waves of 32)
  constexpr int RB = D * 2;        // row bytes (global and LDS)
  constexpr int CPR = D / 8;       // 16-byte chunks per row
  constexpr int KS = D / 32;       // 32-wide k-steps of the score product
  constexpr int DT = D / 16;       // 16-wide d tiles of O^T
  constexpr int BNK = (D == 128) ? FA16_BN128 : BN;  // keys per staged tile
  constexpr int KT = BNK / 16;     // 16-key tiles per KV tile
  constexpr int NKP = KT / 2;      // 32-key k-steps of the PV product
  constexpr int TILE = BNK * RB;   // bytes of one K (or V) tile
  static_assert(D == 64 || D == 128, "head dims of the 16x16x32 kernel");

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  // SUB = 2: a staged unit is a PAIR of tiles (one barrier and one staging pass per 128 keys; slots 2 buf + sub) -- affordable where the
  // workgroups are 256 rows high (two per CU: 2 x 64 KiB of LDS)
  lds_char *Kbuf = smem;                   // [2 SUB][BN][RB], chunks swizzled
  lds_char *Vbuf = smem + 2 * SUB * TILE;  // [2 SUB][BN][RB], chunks swizzled

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int c = lane & 15;  // query within a query tile / key row within a key tile / d within a d tile
  const int g = lane >> 4;  // 16-lane group

  int bh, qb;
  map_block<CAUSAL>(blockIdx.x, p, bh, qb);
  long long base, base_kv;
  head_bases(bh, p, base, base_kv);
  const int coff = p.Nk - p.N;  // causal, Nq != Nk: bottom-right aligned (key j visible to query i iff j <= i + coff)
  const int q0 = qb * BMR;
  const int qw0 = q0 + wave * WM;

  const int GRB = PAD ? p.D * 2 : RB;  // row bytes in global memory
  auto gcol = [&](int chunk) -> unsigned { return (!PAD || chunk * 8 < p.D) ? (unsigned)chunk * 16 : 0x80000000u; };  // (heads below 2 GiB: fa_fwd)
  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.q + base * 2), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.k + base_kv * 2), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.v + base_kv * 2), 0, kv_head_bytes, 0x00020000);

  // ---- Q fragments (B operand of K.Q^T): lane (c, g) holds Q[qw0 + 16qt + c][32ks + 8g .. +7]; rows >= N read as zero
  vec8 qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)(qw0 + 16 * qt + c) * GRB + gcol(4 * ks + g), 0, 0);
      qf[qt][ks] = __builtin_bit_cast(vec8, t);
    }

  // ---- per-lane LDS addresses (absolute, opaque to hipcc: the dynamic-LDS base is a link-time constant it cannot fold)
  // K: ds_read_b128 of row (16kt + c), chunk (4ks + g); the swizzle depends on c only
  const int kx = (D == 64) ? ((c >> 1) & 7) : c;
  const lds_char *kptr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kptr[ks] = Kbuf + c * RB + (((4 * ks + g) ^ kx) << 4);
    asm volatile("" : "+v"(kptr[ks]));
  }
  // V: transposed read; lane 4q + pp of a 16-lane group addresses row (.. + 4g + q), columns 16dt + 4pp .. +3
  const int vq = c >> 2, vp = c & 3;
  const int vrow = 4 * g + vq;
  const int vx = (D == 64) ? (((vrow >> 1) & 3) << 1) : ((vrow & 7) << 1);
  const lds_char *vptr[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    vptr[dt] = Vbuf + vrow * RB + ((((2 * dt) ^ vx) + (vp >> 1)) << 4) + 8 * (vp & 1);
    asm volatile("" : "+v"(vptr[dt]));
  }

  const int kv_end = CAUSAL ? min(p.Nk, q0 + BMR + coff) : p.Nk;
  const int nT = (kv_end + BNK - 1) / BNK;

  // ---- LDS-DMA staging: wave w moves the 1-KiB pieces w, w + 4, ... of each tile; inside a piece the LDS image is
  // lane-linear, so the chunk swizzle sits on the SOURCE address (one per-lane offset for K, one for V)
  constexpr int RPP = 1024 / RB;        // rows per piece
  constexpr int NPW = (BNK / RPP) / RW;  // pieces per wave, tile and operand
  static_assert((RW * RPP) % 16 == 0, "the piece stride must keep the swizzle");
  unsigned dma_kvo, dma_vvo;
  {
    const int row = wave * RPP + lane / CPR, pc = lane % CPR;
    const int skx = (D == 64) ? ((row >> 1) & 7) : (row & 15);
    const int svx = (D == 64) ? (((row >> 1) & 3) << 1) : ((row & 7) << 1);
    dma_kvo = (unsigned)(row * GRB) + gcol(pc ^ skx);
    dma_vvo = (unsigned)(row * GRB) + gcol(pc ^ svx);
  }
  auto stage_dma = [&](int t, int buf) {  // tile t -> buffer buf (hipcc does not count these loads: the caller waits vmcnt(0))
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const unsigned soff = PAD ? (unsigned)(t * BNK + j * RW * RPP) * GRB : (unsigned)t * TILE + j * (RW * 1024);
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)Kbuf + buf * TILE + (wave + RW * j) * 1024;
      const unsigned lv = (unsigned)(__UINTPTR_TYPE__)Vbuf + buf * TILE + (wave + RW * j) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(dma_kvo), "s"(rk), "s"(soff) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lv), "v"(dma_vvo), "s"(rv), "s"(soff) : "memory");
    }
  };

  f32x4 oacc[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int i = 0; i < 4; ++i) oacc[dt][qt][i] = 0.0f;
  constexpr bool ONES = (FA16_ONES != 0);
  // ONES: probabilities are formed against a reference BIAS log2 units ABOVE the row maximum found when the reference was last set
  // (P' = P / 2^BIAS <= 2^-BIAS then), so that "some score has since risen THR = BIAS + 1 above that maximum" reads "some P' >= 2",
  // i.e. "the top exponent bit of some packed P' is set" -- an OR over the 16 packed registers instead of 32 additions. bf16 keeps
  // fp32's exponent range (BIAS 7: THR 8 as before); f16 loses probabilities below 2^-24, so its reference sits only 3 above (THR 4).
  constexpr float BIAS = !ONES ? 0.0f : std::is_same<Tag, F16>::value ? 3.0f : 7.0f;
  // (the reference of row (16qt + c) itself -- log2 units: a stale row max + BIAS -- lives negated in negm[qt] below)
  float l[2] = {0.0f, 0.0f};            // !ONES: this lane's share of the row sums
  f32x4 lacc[2];                        // ONES: the row sums, complete, in every register of the tuple (accumulator of the ones tile)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < 4; ++i) lacc[qt][i] = 0.0f;
  vec8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (elem)1.0f;
  if constexpr (ONES) asm volatile("" : "+v"(ones));
  const float c2 = p.scale * 1.4426950408889634f;  // scale * log2(e)
  // minus the reference in the 4 registers of a tuple per query tile = the C operand of each score chain's first MFMA. It starts at
  // BIAS (ONES: an assumed row maximum of 0, see tile()) / at -inf (sums by v_add: the first tile sets it from the true maxima)
  f32x4 negm[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) negm[qt][i] = ONES ? -BIAS : INFINITY;
    asm volatile("" : "+v"(negm[qt]));  // opaque: else hipcc re-materialises the splat in front of every MFMA
  }

  stage_dma(0, 0);
  if constexpr (SUB == 2) {
    if (1 < nT) stage_dma(1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // Q~ = round(c.Q), once per block; retire the Q loads HERE (hipcc otherwise carries them into the loop as "possibly pending")
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (elem)((float)qf[qt][ks][j] * c2);
      asm volatile("" : "+v"(qf[qt][ks]));
    }
  __syncthreads();

  const float sum_thr = __builtin_exp2f(FA16_DEFER_THR);
  // element (kt, i) of query tile qt is masked iff 16kt + i > lim[qt] (key > query + coff, or key >= Nk); NKT key tiles from tile K0 on
  auto apply_mask = [&](auto k0c, auto &s, const int kv0) __attribute__((always_inline)) {
    constexpr int K0 = decltype(k0c)::value;
    constexpr int NKT = sizeof(s) / sizeof(s[0]);
    int g4 = 4 * g;
    asm volatile("" : "+v"(g4));  // pins the limits and the 32 compares inside the caller's `if (need_mask)`: hipcc otherwise hoists them in front of EVERY tile
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      int lim = p.Nk - 1 - kv0 - g4;
      if (CAUSAL) lim = min(lim, qw0 + 16 * qt + c + coff - kv0 - g4);
#pragma unroll
      for (int k2 = 0; k2 < NKT; ++k2)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[k2][qt][i] = (16 * (K0 + k2) + i > lim) ? -INFINITY : s[k2][qt][i];
    }
  };
  // mask only on tiles that cross the diagonal or the end of the sequence (re-evaluated from scalars at every use: carried as a
  // bool across the passes of a tile hipcc kept it in a VGPR, v_cndmask + v_cmp per tile)
  auto needs_mask = [&](const int kv0) __attribute__((always_inline)) { return (CAUSAL && (kv0 + BNK - 1 > qw0 + coff)) || (kv0 + BNK > p.Nk); };
  // S^T = K.Q^T + C for NKT key tiles from tile K0 on: s[k2][qt][i] = S[query 16qt + c][key kv0 + 16(K0 + k2) + 4g + i] + c_[qt];
  // K fragments read LA ahead of their use
  auto score_group = [&](auto bufc, auto k0c, auto &s, const f32x4 c0, const f32x4 c1) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    constexpr int K0 = decltype(k0c)::value;
    constexpr int NKT = sizeof(s) / sizeof(s[0]);
    constexpr int NK = NKT * KS, LA = (FA16_LAK < NK) ? FA16_LAK : NK;
    vec8 kf[NK];
    auto kread = [&](int i) { kf[i] = __builtin_bit_cast(vec8, lds_read_b128(kptr[i % KS] + buf * TILE + (K0 + i / KS) * 16 * RB)); };
#pragma unroll
    for (int i = 0; i < LA; ++i) kread(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int k2 = i / KS, ks = i % KS;
      s[k2][0] = M::mfma(kf[i], qf[0][ks], ks == 0 ? c0 : s[k2][0]);
      s[k2][1] = M::mfma(kf[i], qf[1][ks], ks == 0 ? c1 : s[k2][1]);
      if (i + LA < NK) kread(i + LA);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // pack the probabilities of NKT key tiles (from K0 on) into the PV operands pf[kp][qt] (k-step kp = two key tiles), rounded to
  // the input type; ONES: OR the packed words into `bits`
  auto pack_group = [&](auto k0c, auto &s, vec8 (&pf)[NKP][2], unsigned &bits) __attribute__((always_inline)) {
    constexpr int K0 = decltype(k0c)::value;
    constexpr int NKT = sizeof(s) / sizeof(s[0]);
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[K0 / 2 + kp][qt][j] = (elem)s[2 * kp + (j >> 2)][qt][j & 3];
        if constexpr (ONES) {
          const u32x4 w = __builtin_bit_cast(u32x4, pf[K0 / 2 + kp][qt]);
          bits |= w[0] | w[1] | w[2] | w[3];
        }
      }
  };
  // the hot pass over NKT key tiles: P = exp2(S') -> pf (and `bits`, or the row sums ls)
  auto hot_group = [&](auto bufc, auto k0c, auto nktc, const int kv0, vec8 (&pf)[NKP][2], unsigned &bits,
                       float (&ls)[2]) __attribute__((always_inline)) {
    constexpr int NKT = decltype(nktc)::value;
    f32x4 s[NKT][2];
    score_group(bufc, k0c, s, negm[0], negm[1]);
    if (needs_mask(kv0)) apply_mask(k0c, s, kv0);
#if FA16_PRIO
    if constexpr (decltype(k0c)::value + NKT == KT) __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int k2 = 0; k2 < NKT; ++k2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          s[k2][qt][i] = __builtin_amdgcn_exp2f(s[k2][qt][i]);
          if constexpr (!ONES) ls[qt] = (decltype(k0c)::value == 0 && k2 == 0 && i == 0) ? s[k2][qt][i] : ls[qt] + s[k2][qt][i];
        }
    pack_group(k0c, s, pf, bits);
  };
  // the row maxima of the raw scores of NKT key tiles, into mx[qt] (this lane's keys)
  auto max_group = [&](auto bufc, auto k0c, auto nktc, const int kv0, float (&mx)[2]) __attribute__((always_inline)) {
    constexpr int NKT = decltype(nktc)::value;
    f32x4 s[NKT][2], zero;
#pragma unroll
    for (int i = 0; i < 4; ++i) zero[i] = 0.0f;
    score_group(bufc, k0c, s, zero, zero);
    if (needs_mask(kv0)) apply_mask(k0c, s, kv0);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int k2 = 0; k2 < NKT; ++k2)
#pragma unroll
        for (int i = 0; i < 4; ++i) mx[qt] = fmaxf(mx[qt], s[k2][qt][i]);
  };
  // row maxima of the raw scores -> new reference: rescale O and the row sums, rewrite the C-operand tuples (first tile and rare path)
  auto new_reference = [&](auto firstc, float (&mx)[2]) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(firstc)::value;  // first tile: O and the row sums are still zero, nothing to rescale
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float v = mx[qt];
      v = fmaxf(v, xlane(v, 16));
      v = fmaxf(v, xlane(v, 32));
      const float m_old = -negm[qt][0];
      const float m_new = FIRST ? v + BIAS : fmaxf(m_old, v + BIAS);  // finite: every row sees key 0 of the first tile
#pragma unroll
      for (int i = 0; i < 4; ++i) negm[qt][i] = -m_new;
      asm volatile("" : "+v"(negm[qt]));
      if constexpr (FIRST) continue;
      const float alpha = __builtin_amdgcn_exp2f(m_old - m_new);
      if constexpr (ONES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lacc[qt][i] *= alpha;
      } else {
        l[qt] *= alpha;
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[dt][qt][i] *= alpha;
    }
  };
  constexpr int G = (FA16_HALVES != 0) ? 2 : 1;  // key-tile groups of the hot pass: 1 = all 16 score MFMAs, then the softmax; 2 = per 32-key half
  using KTG = std::integral_constant<int, KT / G>;
  using K0A = std::integral_constant<int, 0>;
  using K0B = std::integral_constant<int, KT / 2>;

  // One KV tile: P = exp2(S') against the running reference (S' comes out of the matrix core with the reference already subtracted), and
  // a test whether the reference is stale -- ONES: some packed P' has its top exponent bit set (P' >= 2, inf or NaN); else: a lane's 16
  // probabilities of a row add up to more than 2^THR. Only then -- and on the first tile, whose reference is -inf -- the wave takes the
  // rare path: one pass for the row maxima of the raw scores, rescale O and the row sums, and the hot pass AGAIN (a backward branch:
  // the score / P registers of the hot pass never merge with anything the rare path computes -- with the rare path as a forward
  // if / else hipcc parked all 32 P values in scratch on the common path of the non-causal kernel).
  auto tile = [&](auto bufc, auto firstc, const int t) {
    constexpr int buf = decltype(bufc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    const int kv0 = t * BNK;

    // whole-tile skip per wave (kernels.metal:682 with Br = 32)
    // (non-causal: an always-true scalar hipcc cannot see through -- with the tile body unconditional it schedules across the
    // tile boundary and spills 250-390 B per lane under the 128-register budget; with the branch in place, as in the causal
    // kernel, it fits)
    int always = 1;
    if constexpr (!CAUSAL) asm volatile("" : "+s"(always));
    const bool wave_active = CAUSAL ? (kv0 <= qw0 + WM - 1 + coff) : (always != 0);
    if (wave_active) {
      vec8 pf[NKP][2];  // B operands of the PV product: k-step kp, query tile qt
      float ls[2] = {0.0f, 0.0f};
      // redo (wave-uniform): first the row maxima of the raw scores -> new reference, then the hot pass. The FIRST tile starts without
      // it (ONES): its reference is the constant BIAS, i.e. an assumed row maximum of 0 -- any finite reference is as good as the true
      // maximum as long as no P' reaches 2 (the hot pass's own test) and the row sums do not vanish (tested once, behind the first
      // tile's PV product: scores below -2^6 only) -- which saves every block a whole score pass (16 MFMAs, the maxima, their cross-lane
      // steps). (A 16-key reference for the first tile, 4 MFMAs instead of 16, was measured: no gain, profiles/r04/ab_mfma16_first_tile.log.)
      bool redo = FIRST && !ONES;
      for (;;) {
        if (__builtin_expect(redo, 0)) {
          float mx[2] = {-INFINITY, -INFINITY};
          max_group(bufc, K0A{}, std::integral_constant<int, KT / 2>{}, kv0, mx);
          max_group(bufc, K0B{}, std::integral_constant<int, KT / 2>{}, kv0, mx);
          new_reference(firstc, mx);
        }
#if FA16_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        unsigned bits = 0;
        hot_group(bufc, K0A{}, KTG{}, kv0, pf, bits, ls);
        if constexpr (G == 2) hot_group(bufc, K0B{}, KTG{}, kv0, pf, bits, ls);
        bool stale;
        if constexpr (ONES) stale = __builtin_amdgcn_ballot_w64((bits & 0x40004000u) != 0) != 0;
        else stale = __builtin_amdgcn_ballot_w64(fmaxf(ls[0], ls[1]) > sum_thr) != 0;
        if (__builtin_expect(stale && !redo, 0)) {  // (after a redo every P' <= 2^-BIAS: a second stale reading is inf / NaN input)
          redo = true;
          continue;
        }
        if constexpr (!ONES) {
          l[0] += ls[0];
          l[1] += ls[1];
        }
#if FA16_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        // ---- O^T += V^T.P^T, and (ONES) the row sums from a fifth d tile of ones
        {
          constexpr int NV = NKP * DT, LA = FA16_LAV;
          s16x4 wlo[NV], whi[NV];
          auto vread = [&](int j) {  // step j = (kp, dt)
            const lds_char *vb = vptr[j % DT] + buf * TILE + (32 * (j / DT)) * RB;
            wlo[j] = lds_read_tr16(vb);            // keys 32kp + 4g + 0..3       (k elements 0..3)
            whi[j] = lds_read_tr16(vb + 16 * RB);  // keys 32kp + 16 + 4g + 0..3  (k elements 4..7)
          };
#pragma unroll
          for (int j = 0; j < LA; ++j) vread(j);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const s16x8 v8 = __builtin_shufflevector(wlo[j], whi[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) oacc[j % DT][qt] = M::mfma(__builtin_bit_cast(vec8, v8), pf[j / DT][qt], oacc[j % DT][qt]);
            if (j + LA < NV) vread(j + LA);
            if constexpr (ONES) {
              if (j % DT == DT - 1) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) lacc[qt] = M::mfma(ones, pf[j / DT][qt], lacc[qt]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#if FA16_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if constexpr (FIRST && ONES) {
          // the assumed reference was too HIGH for some row (every score of its first 64 keys below about -2^6: sums under 2^-64):
          // start the tile over with the true maxima (O and the row sums hold nothing of weight: cleared)
          if (__builtin_expect(!redo && __builtin_amdgcn_ballot_w64(fminf(lacc[0][0], lacc[1][0]) < 0x1p-64f) != 0, 0)) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
              for (int i = 0; i < 4; ++i) lacc[qt][i] = 0.0f;
#pragma unroll
              for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) oacc[dt][qt][i] = 0.0f;
            }
            redo = true;
            continue;
          }
        }
        break;
      }
    }
  };
  // one staged unit (SUB tiles from tile t0 on, in the slots of buffer `buf`): the next unit's tiles go into the other buffer first, in
  // flight under this unit's MFMAs; one wait + barrier at its end
  auto unit = [&](auto bufc, auto firstc, const int t0) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub)
      if (t0 + SUB + sub < nT) stage_dma(t0 + SUB + sub, (buf ^ 1) * SUB + sub);
    tile(std::integral_constant<int, buf * SUB>{}, firstc, t0);
    if constexpr (SUB == 2) {
      if (t0 + 1 < nT) tile(std::integral_constant<int, buf * SUB + 1>{}, std::false_type{}, t0 + 1);
    }
    if (t0 + SUB < nT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces issued at the top of this unit have landed
    __syncthreads();
  };
  unit(std::integral_constant<int, 0>{}, std::true_type{}, 0);
  for (int t = SUB; t < nT; t += 2 * SUB) {
    unit(std::integral_constant<int, 1>{}, std::false_type{}, t);
    if (t + SUB < nT) unit(std::integral_constant<int, 0>{}, std::false_type{}, t + SUB);
  }

  // ---- epilogue: normalise, LSE, O tile -> LDS -> coalesced 16-byte stores
  lds_char *Ot = smem + wave * (WM * RB);  // this wave's [32][D] tile (inside the K buffers: free since the last barrier)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = ONES ? lacc[qt][0] : l[qt];
    if constexpr (!ONES) {
      lt += xlane(lt, 16);
      lt += xlane(lt, 32);
    }
    const float inv_l = __builtin_amdgcn_rcpf(lt);  // v_rcp_f32 (1 ulp): the IEEE division costs ten instructions per row
    const int row = 16 * qt + c, qrow = qw0 + row;
    // LSE = (reference + log2 l) . ln 2, v_log_f32 straight (l is a sum of probabilities around 2^-BIAS .. 2^THR: no denormals)
    if (p.lse != nullptr && g == 0 && qrow < p.N)
      p.lse[(long long)bh * p.N + qrow] = (__builtin_amdgcn_logf(lt) - negm[qt][0]) * 0.6931471805599453f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      // registers 0..3 = d columns 16dt + 4g + 0..3 of row (16qt + c); converted as one vector so that hipcc packs pairs
      typedef elem elem4 __attribute__((ext_vector_type(4)));
      elem4 e;
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (elem)(oacc[dt][qt][i] * inv_l);
      const u32x2 w = __builtin_bit_cast(u32x2, e);
      const int ch = (2 * dt + (g >> 1)) ^ (row & (CPR - 1));  // chunk XOR row spreads the rows over the banks
      lds_write_b64(Ot + row * RB + (ch << 4) + 8 * (g & 1), w);
    }
  }
  __syncthreads();
  {
    elem *Og = (elem *)p.o + base;
#pragma unroll
    for (int it = 0; it < WM * CPR / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR, ch = idx % CPR;
      const u32x4 vv = lds_read_b128(Ot + row * RB + ((ch ^ (row & (CPR - 1))) << 4));
      if (qw0 + row < p.N && (!PAD || ch * 8 < p.D)) *reinterpret_cast<u32x4 *>(Og + (long long)(qw0 + row) * (PAD ? p.D : D) + ch * 8) = vv;
    }
  }
}

template <typename Tag, int D, bool CAUSAL, int RW, bool PAD = false, int SUB = 1>
__global__ __launch_bounds__(64 * RW, (D == 64 ? FA16_OCC : FA16_OCC128)) void fwd_mfma16_kernel(Params p) {
  fwd_mfma16_body<Tag, D, CAUSAL, RW, PAD, SUB>(p);
}

bool mfma16_supported(int dtype, int D) { return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) && D >= 8 && D <= 128 && D % 8 == 0; }

template <typename Tag, int D, bool CAUSAL, int RW, bool PAD = false, int SUB = 1>
static hipError_t launch16_one(const Params &p, hipStream_t s) {
  const int nQ = (p.N + RW * WM - 1) / (RW * WM);
  const size_t smem = 4 * SUB * (size_t)(D == 128 ? FA16_BN128 : BN) * D * 2;
  auto kern = fwd_mfma16_kernel<Tag, D, CAUSAL, RW, PAD, SUB>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem);
    if (e != hipSuccess) return e;
  }
  Params pp = p;
  pp.head_group = causal_head_group(p, D, 2);
#ifdef FA16_FORCE_HEAD_GROUP  // scheduling experiments only (tools/ab.py arms): never defined in the shipped library
  pp.head_group = FA16_FORCE_HEAD_GROUP;
#endif
  set_block_divisors(pp, nQ, pp.head_group);
  (void)hipGetLastError();  // do not report an older sticky error as this launch's
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(64 * RW), smem, s, pp);
  return hipGetLastError();
}

// Waves per workgroup: 4 (128 query rows) or 8 (256 rows, the same 32 rows per wave: eight waves share every K / V tile, so the
// L2 -> LDS stream and its issue slots halve, and a workgroup that is alone on its CU in the launch's tail still runs two waves per SIMD).
// The price is the causal diagonal (a block's lower waves idle through up to seven more tiles' barriers), twice the weight of the
// heaviest block, and a wider spread of launch times on large causal grids. Measured interleaved on warm clocks, medians, bf16 unless
// noted (profiles/r04/ab_mfma16_eight_waves_steady_state.log):
//   non-causal  N = 2048 / 4096 (64 heads) +5.1 / +4.6 %, N = 8192 +4.1 %, 32 heads x 2048 +1.2 % (256 workgroups: level)
//   causal      64 heads x 4096 (config 3) +3.0 %, f16 +3.7 %, x 5120 +1.5 %; but 80 .. 256 heads x 4096 -2.3 .. -6.3 % (best times level),
//               48 heads level, 64 heads x 2048 -4.8 %, x 6144 -0.9 %, x 8192 +0.7 %, x 16384 +0.4 %, 48 heads x 8192 -0.2 %
//   head_dim 128: non-causal 32 heads x 8192 +2.5 %, causal level
// (N = 512, 256 / 512 heads: +3.4 % over four waves) -> eight waves for non-causal grids of at least 512 such workgroups, and under the mask only for the two-round grids of config 3's
// kind (1024 .. 1279 workgroups of 256 rows at 4096 <= N < 6144), where the launch is tail-dominated.
int mfma16_waves(int D, int BH, int N, int Nk, int is_causal) {
#ifdef FA16_FORCE_RW
  return FA16_FORCE_RW;
#endif
  if (D != 64 && D != 128) return 4;  // (padded head dims)
  const long long b256 = (long long)BH * ((N + 255) / 256);
  if (D == 128) return (!is_causal && Nk >= 8192 && b256 >= 1024) ? 8 : 4;
  if (!is_causal) return (Nk >= 512 && b256 >= 512) ? 8 : 4;
  return (Nk >= 4096 && Nk < 6144 && b256 >= 1024 && b256 < 1280) ? 8 : 4;
}

hipError_t launch_mfma16(const Params &p, int dtype, hipStream_t s) {
  const bool w8 = mfma16_waves(p.D, p.B * p.H, p.N, p.Nk, p.is_causal) == 8 && (p.D == 64 || !p.is_causal);
  auto go = [&](auto tag) -> hipError_t {
    using Tag = decltype(tag);
    if (p.D != 64 && p.D != 128) {  // zero-padded rows of the next larger instantiation (four waves)
      if (p.D < 64) return p.is_causal ? launch16_one<Tag, 64, true, 4, true>(p, s) : launch16_one<Tag, 64, false, 4, true>(p, s);
      return p.is_causal ? launch16_one<Tag, 128, true, 4, true>(p, s) : launch16_one<Tag, 128, false, 4, true>(p, s);
    }
    if (p.D == 64) {
      if (w8) return p.is_causal ? launch16_one<Tag, 64, true, 8, false, FA16_SUB8>(p, s) : launch16_one<Tag, 64, false, 8, false, FA16_SUB8>(p, s);
      return p.is_causal ? launch16_one<Tag, 64, true, 4>(p, s) : launch16_one<Tag, 64, false, 4>(p, s);
    }
    if (p.D == 128) {
      if (w8) return launch16_one<Tag, 128, false, 8>(p, s);  // (non-causal only: mfma16_waves)
      return p.is_causal ? launch16_one<Tag, 128, true, 4>(p, s) : launch16_one<Tag, 128, false, 4>(p, s);
    }
    return hipErrorInvalidValue;
  };
  return dtype == FA_DTYPE_F16 ? go(F16{}) : go(BF16{});
}

}  // namespace fa
