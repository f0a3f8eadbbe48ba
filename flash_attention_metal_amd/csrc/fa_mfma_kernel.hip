// fa_mfma_kernel.hip -- the operator on the CDNA4 matrix cores.
//
// Replaces /root/reference/kernels.metal:600-883 (flash_attention_v4_half_kernel)
// and :177-455 (flash_attention_simd_kernel): same math -- tiled QK^T ->
// online softmax -> PV, causal predicate `key > query -> masked`
// (kernels.metal:748), whole-tile skip (kernels.metal:682), L = m + ln(l)
// (kernels.metal:862-864) -- nothing else is shared with it. fp32 accumulators
// for S, O, m, l (the reference accumulates S and O in half).
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head)):
//   * each wave owns 32 query rows; its Q fragments stay in registers
//   * K/V tiles of 64 keys are double-buffered in LDS: the next tile's
//     128-bit buffer loads are issued before the current tile's arithmetic and
//     written to the other buffer after it; one barrier per tile
//   * S^T = K.Q^T with v_mfma_f32_32x32x16 ("swapped" product): the query index
//     lands on the lane, the 32 key scores of a block in that lane's 16
//     accumulator registers (+16 in lane^32), so row max / row sum are
//     in-register reductions plus ONE v_permlane32_swap -- no LDS round trip
//   * P^T feeds the PV product straight from those registers as the B operand
//     (k order inside a step is the accumulator's row order; V^T is read from
//     LDS in the same order with ds_read_b64_tr_b16, so V stays row-major)
//   * O^T accumulates in 16*(D/32) registers; the O rescale is skipped when no
//     row max in the wave moved (exact: alpha == 1)
//   * K rows are XOR-swizzled in LDS for conflict-free ds_read_b128, V rows for
//     conflict-free transposed reads
//   * epilogue: O tile -> LDS -> whole 128-byte rows, 16 B per lane
//   * grid: 1-D, heads dealt to XCDs (blocks b and b+8 share an L2) so one
//     head's K/V stays in one L2; causal q-blocks heaviest first
#include <stdlib.h>

#include <algorithm>

#include "fa_mfma_common.h"

#ifndef FA_DEFER_THR
#define FA_DEFER_THR 8.0f  // log2 units: P values are bounded by 2^8 between rescales (0 = rescale whenever a max moved)
#endif
#ifndef FA_SUMTRIG
#define FA_SUMTRIG 1  // 1: the row sums decide whether the reference max is stale (no per-tile max); 0: per-tile max vs threshold
#endif
#ifndef FA_PRESCALE
#define FA_PRESCALE 1  // 1 (f16/bf16): Q~ = round(scale.log2e.Q) once per block and -m (log2 units) as the C operand of each score chain: P = exp2(S') with no FMA
#endif
#ifndef FA_VPRE_HALF
#define FA_VPRE_HALF 0  // 1: only the first 32 keys' V^T fragments are prefetched under the QK^T MFMAs, the second half under the first half's PV MFMAs (-16 live
                        // registers: needed while K/V were staged through registers; with LDS-DMA staging the full prefetch fits -- 162-164 VGPR, no scratch --
                        // and is 0.5-0.9 % faster, profiles/r03/ab_knobs_final_build.log)
#endif
#ifndef FA_MAIN_OCC4
#define FA_MAIN_OCC4 1  // 1: the plain 128-row kernel at head_dim 64 (16-bit inputs, pre-scaled operand) gives up the V^T prefetch below and
                        // fits 128 registers: FOUR workgroups per CU instead of three. Config 3 +2.6 %, N=8192 +2.7 %, non-causal N=4096
                        // +1.6 %, N<=2048 -0.3..-1.7 % (profiles/r03/ab_four_waves_per_simd.log). Round 3 first tried four with the prefetch
                        // in place and spilled 150-210 B; the split / 64-row / exact forms keep the prefetch and three.
#endif
#ifndef FA_MFMA_DMA
#define FA_MFMA_DMA 1  // 1 (f16/bf16, head_dim 32/64/128): K/V tiles go global -> LDS by LDS-DMA (buffer_load ... lds), no staging registers, no
                       // ds_write: config 3 +6.7 %, head_dim 128 +9..11 %, bit-identical outputs (profiles/r03/ab_mfma_lds_dma.log); 0 = register staging
#endif
#ifndef FA_LAK
#define FA_LAK (D == 128 ? 4 : 2)  // head dims other than 64: K fragments are read this many MFMAs ahead of their use (head_dim 128: 4 since the
#endif                             // LDS-DMA staging freed the registers: +1..2 %, profiles/r03/ab_dma_knobs.log)
#ifndef FA_LAV
#define FA_LAV (D == 128 ? 4 : 2)  // ... and V^T fragments
#endif
#ifndef FA_PRIO
#define FA_PRIO 2  // wave priority: 2 = raised around the MFMA clusters (+0.4..0.9 % A/B), 1 = around the softmax (-1..-6 %), 0 = off
#endif

namespace fa {

// Which (dtype, head_dim) the pre-scaled query operand exists for: fp8 Q cannot carry the factor (3 mantissa bits), and
// head_dim 256 has no 16 registers to spare for the row-constant tuple.
template <typename Tag, int D>
constexpr bool prescale_applies() {
  return (FA_PRESCALE != 0) && !std::is_same<Tag, FP8>::value && D <= 128;
}

// head dims: 32, 64, 96, 128, 256 (scope row f3). LDS rows keep a power-of-two pitch (head_dim 96 rows sit in
// 256-byte slots) so the XOR swizzles stay inside a row; head_dim 256 needs the whole register file
// (128 accumulators for O^T alone): one workgroup per CU there.
// SPLIT = 2 ("split2" kernel, for grids that leave most of the chip idle): eight waves per workgroup, waves 0-3 and 4-7
// take the even and the odd KV tiles of the same 128 query rows -- each half with its own K/V buffers, staging and
// (m, l, O^T) -- and merge once through LDS by their reference maxima. Halves the sequential tile count of a block.
// PRESC: the pre-scaled query operand (see PRE below); false = every score scaled in fp32 (variant mfma_exact).
// ROWS: query rows per workgroup, 128 (four row groups of 32 = four waves per split) or 64 (two): "h64s2" = 64 rows x 2 splits
// is a four-wave workgroup whose wave pairs take the even / odd KV tiles -- twice the workgroups and half the sequential
// tiles of a block, for causal grids whose critical path is the heaviest q block (N <= 2048 at 64 heads).
// the instantiations of the plain kernel that run four workgroups per CU (FA_MAIN_OCC4)
template <typename Tag, int D, bool PRESC>
constexpr bool main_kernel_occ4() {
  return (FA_MAIN_OCC4 != 0) && D == 64 && PRESC && prescale_applies<Tag, D>();
}

template <typename Tag, int D, bool CAUSAL, int SPLIT, bool PRESC, int ROWS = BM>
__device__ __forceinline__ void fwd_mfma_body(const Params &p) {
  constexpr int RW = ROWS / WM;   // row groups = waves per split
  constexpr int ST = 64 * RW;     // threads per split (the staging map's width)
  using M = MT<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  constexpr int GRB16 = D * 2;              // row bytes of a 16-bit row in global memory
  constexpr int RB = (D == 96) ? 256 : D * 2;  // LDS row pitch (bytes)
  constexpr int CPR = D / 8;                // 16-byte chunks per row that hold data
  constexpr int CPRL = RB / 16;             // 16-byte slots per LDS row (power of two)
  constexpr int KS = D / 16;                // k-steps of the QK^T product
  constexpr int DB = D / 32;                // 32-wide d blocks of O^T
  constexpr int TILE = BN * RB;             // bytes of one K (or V) tile in LDS
  constexpr bool IS_FP8 = std::is_same<Tag, FP8>::value;  // Q,K,V are e4m3 in HBM, bf16 from LDS onwards
  constexpr int GB = IS_FP8 ? 1 : 2;        // bytes per element in HBM
  constexpr int GRB = D * GB;               // global row bytes
  constexpr int GTILE = BN * GRB;           // global bytes of one K (or V) tile
  constexpr int NCH = BN * (GRB / 16) / ST;  // staged 16-byte global chunks per thread per tile
  constexpr bool MAIN4 = (FA_MAIN_OCC4 != 0) && main_kernel_occ4<Tag, D, PRESC>() && SPLIT == 1 && ROWS == BM;
  constexpr bool VPRE = (D == 64) && !IS_FP8 && !MAIN4;  // prefetch V^T fragments under the QK^T MFMAs
  // Pre-scaled operand (f16/bf16): the Q fragments are multiplied by c = scale.log2(e) and rounded back to the input
  // type ONCE per block, and the running reference -m (log2 units) is the C operand of the first MFMA of every score
  // chain, so the matrix core hands out S' = c.q.k - m and P = exp2(S') needs no v_fma (kernels.metal:763-771 scales
  // and subtracts per score). Costs one rounding of c.q (2^-9 relative for bf16, 2^-12 for f16) per operand element:
  // the score error stays a fraction of the P rounding that follows; LSE error scales with the score magnitude
  // (include/fa_mi355.h states the bound).
  constexpr bool PRE = PRESC && prescale_applies<Tag, D>();
  // fp8 inputs: the score product runs on v_mfma_scale_f32_32x32x64_f8f6f4 (unit E8M0 scales: an exact e4m3 product
  // at twice the bf16 rate, K = 64 per instruction). K stays e4m3 in LDS (rows of D bytes) and Q stays e4m3 in
  // registers; V is widened to bf16 while it is staged, because P has to be bf16 for the PV product anyway.
  constexpr int KRB = IS_FP8 ? D : RB;      // K row pitch in LDS (bytes)
  constexpr int KTILE = BN * KRB;           // bytes of one K tile in LDS
  constexpr int NS8 = IS_FP8 ? D / 64 : 1;  // fp8: 64-wide k-steps of the score product
  typedef int i32x8 __attribute__((ext_vector_type(8)));

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int sp = (SPLIT == 1) ? 0 : (wave_all / RW);  // which KV split this wave works on
  constexpr int GROUP_LDS = 2 * KTILE + 2 * TILE;     // K and V double buffers of one split
  lds_char *smem = (lds_char *)smem_generic + sp * GROUP_LDS;
  lds_char *Kbuf = smem;              // [2][BN][KRB], rows swizzled
  lds_char *Vbuf = smem + 2 * KTILE;  // [2][BN][RB], rows swizzled

  const int tid = threadIdx.x & (ST - 1);  // thread within its split's waves (staging map)
  const int lane = tid & 63;
  const int wave = wave_all & (RW - 1);          // row group: query rows 32*wave .. +31 of the block
  const int r = lane & 31;  // query within the wave / key row within a block
  const int h = lane >> 5;  // lane half

  // ---- block -> (batch*head, q block). blocks b and b+8 share an XCD's L2:
  // deal heads to the 8 residues so a head's K/V stays in one L2.
  int bh, qb;
  map_block<CAUSAL>(blockIdx.x, p, bh, qb);
  long long base, base_kv;
  head_bases(bh, p, base, base_kv);
  // grouped-query heads: query head h reads key/value head h / (H / Hkv); Nk keys per head.
  // Causal with Nq != Nk is bottom-right aligned: key j visible to query i iff j <= i + coff.
  const int coff = p.Nk - p.N;
  const int q0 = qb * ROWS;
  const int qw0 = q0 + wave * WM;  // first query row of this wave
  const int qrow = qw0 + r;

  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
      (void *)((const char *)p.q + base * GB), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
      (void *)((const char *)p.k + base_kv * GB), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      (void *)((const char *)p.v + base_kv * GB), 0, kv_head_bytes, 0x00020000);

  // ---- Q fragments (B operand of K.Q^T): lane (r,h) holds Q[qrow][16ks+8h .. +7].
  // Rows >= N read as zero through the descriptor's range check.
  vec8 qf[IS_FP8 ? 1 : KS];
  i32x8 qf8[NS8];  // fp8: k-step j holds Q[qrow][64j + 32h .. +31] (32 e4m3 values; any k order works as long as K uses the same)
  if constexpr (IS_FP8) {
#pragma unroll
    for (int j = 0; j < NS8; ++j) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * GRB + 64 * j + 32 * h, 0, 0);
      const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * GRB + 64 * j + 32 * h + 16, 0, 0);
      qf8[j] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * GRB16 + (2 * ks + h) * 16, 0, 0);
      qf[ks] = __builtin_bit_cast(vec8, t);
    }
  }

  // ---- per-lane LDS offsets
  // K: ds_read_b128 of row (32kb + r), chunk (2ks + h); swizzle depends on r only
  const int kx = (D == 32) ? ((r >> 2) & 3) : (D == 64) ? ((r >> 1) & 7) : (r & 15);
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = r * RB + (((2 * ks + h) ^ kx) << 4);
  // fp8 K rows are D bytes: the row image equals a 16-bit row of head_dim D/2 (same swizzle family)
  const int kx8 = (D == 64) ? ((r >> 2) & 3) : (D == 128) ? ((r >> 1) & 7) : (r & 15);
  // V: transposed read; 16-lane group g covers d columns 16(g&1).. of block db,
  // lane 4q+pp of the group addresses row (.. + 4h + q), columns 4pp..4pp+3
  const int g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  const int vx = (D == 32) ? 0 : (D == 64) ? (((vq >> 1) & 1) << 2) : (vq << 2);
  int voff[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
    voff[db] = (4 * h + vq) * RB + ((((4 * db) ^ vx) + 2 * g1 + (vp >> 1)) << 4) + 8 * (vp & 1);

  // Absolute LDS addresses of this lane's fragment reads, made opaque ONCE: the dynamic-LDS base is a link-time
  // constant hipcc cannot fold, and with plain offsets it re-added it ("v_add_u32 x, 0, y") six times per tile.
  const lds_char *kptr[KS], *vptr[DB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kptr[ks] = Kbuf + koff[ks];
    asm volatile("" : "+v"(kptr[ks]));
  }
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    vptr[db] = Vbuf + voff[db];
    asm volatile("" : "+v"(vptr[db]));
  }

  // ---- staging: thread -> NCH 16-byte chunks of the K tile and of the V tile
  // (fp8: a 16-byte global chunk holds 16 elements = chunks 2c and 2c+1 of the bf16 row image in LDS,
  //  each swizzled on its own: st_k/st_v address chunk 2c, st_k1/st_v1 chunk 2c+1)
  constexpr int GCPR = GRB / 16;  // global chunks per row
  int st_g[NCH], st_k[NCH], st_v[NCH], st_k1[IS_FP8 ? NCH : 1], st_v1[IS_FP8 ? NCH : 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * ST;
    const int row = c / GCPR, gch = c % GCPR;
    st_g[i] = row * GRB + gch * 16;
    const int skx = (D == 32) ? ((row >> 2) & 3) : (D == 64) ? ((row >> 1) & 7) : (row & 15);
    const int svx = (D == 32) ? 0 : (D == 64) ? (((row >> 1) & 1) << 2) : ((row & 3) << 2);
    const int ch = IS_FP8 ? 2 * gch : gch;
    st_v[i] = row * RB + ((ch ^ svx) << 4);
    if constexpr (IS_FP8) {
      const int skx8 = (D == 64) ? ((row >> 2) & 3) : (D == 128) ? ((row >> 1) & 7) : (row & 15);
      st_k[i] = row * KRB + ((gch ^ skx8) << 4);  // raw e4m3 chunk
      st_k1[i] = 0;
      st_v1[i] = row * RB + (((ch + 1) ^ svx) << 4);
    } else {
      st_k[i] = row * RB + ((ch ^ skx) << 4);
    }
  }

  const int kv_end = CAUSAL ? min(p.Nk, q0 + ROWS + coff) : p.Nk;
  const int nT = (kv_end + BN - 1) / BN;

  // LDS-DMA staging (FA_MFMA_DMA): wave w of a split moves the 1-KiB pieces w, w + RW, ... of each tile; inside a piece the
  // LDS image is lane-linear, so the chunk swizzle sits on the SOURCE address (one per-lane offset for K, one for V)
  constexpr bool DMA = (FA_MFMA_DMA != 0) && !IS_FP8 && (D == 32 || D == 64 || D == 128);
  constexpr int RPP = 1024 / RB;                 // rows per piece
  constexpr int NPW = (BN / RPP) / RW;           // pieces per wave, tile and operand
  unsigned dma_kvo = 0, dma_vvo = 0;
  if constexpr (DMA) {
    static_assert((RW * RPP) % 16 == 0 || D == 32, "the piece stride must keep the swizzle");
    const int row = wave * RPP + lane / CPRL, pc = lane % CPRL;
    const int skx = (D == 32) ? ((row >> 2) & 3) : (D == 64) ? ((row >> 1) & 7) : (row & 15);
    const int svx = (D == 32) ? 0 : (D == 64) ? (((row >> 1) & 1) << 2) : ((row & 3) << 2);
    dma_kvo = (unsigned)(row * GRB + ((pc ^ skx) << 4));
    dma_vvo = (unsigned)(row * GRB + ((pc ^ svx) << 4));
  }
  auto stage_dma = [&](int t, int buf) {  // tile t -> buffer buf (hipcc does not count these loads: the caller waits vmcnt(0))
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const unsigned soff = (unsigned)t * GTILE + j * (RW * 1024);
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)Kbuf + buf * KTILE + (wave + RW * j) * 1024;
      const unsigned lv = (unsigned)(__UINTPTR_TYPE__)Vbuf + buf * TILE + (wave + RW * j) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(dma_kvo), "s"(rk), "s"(soff) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lv), "v"(dma_vvo), "s"(rv), "s"(soff) : "memory");
    }
  };
  // fp8 inputs: the K tile stays raw e4m3 in LDS (rows of D bytes), so IT can travel by LDS-DMA; V is widened to bf16 on the way
  // and keeps the register path
  constexpr bool DMA_K8 = (FA_MFMA_DMA != 0) && IS_FP8;
  constexpr int RPP8 = 1024 / KRB, NPW8 = (BN / RPP8) / RW;  // rows per piece / pieces per wave of an e4m3 K tile
  unsigned dma_k8o = 0;
  if constexpr (DMA_K8) {
    static_assert((RW * RPP8) % 16 == 0 && NPW8 >= 1, "the piece stride must keep the swizzle");
    constexpr int CPR8 = KRB / 16;
    const int row = wave * RPP8 + lane / CPR8, pc = lane % CPR8;
    const int skx8 = (D == 64) ? ((row >> 2) & 3) : (D == 128) ? ((row >> 1) & 7) : (row & 15);
    dma_k8o = (unsigned)(row * GRB + ((pc ^ skx8) << 4));
  }
  auto stage_dma_k8 = [&](int t, int buf) {
#pragma unroll
    for (int j = 0; j < NPW8; ++j) {
      const unsigned soff = (unsigned)t * GTILE + j * (RW * 1024);
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)Kbuf + buf * KTILE + (wave + RW * j) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(dma_k8o), "s"(rk), "s"(soff) : "memory");
    }
  };
  u32x4 kst[NCH], vst[NCH];
  auto stage_load = [&](int t) {
    const unsigned g0 = (unsigned)t * GTILE;  // tile t starts at key t*BN
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if constexpr (!DMA_K8) kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, g0 + st_g[i], 0, 0);
      vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, g0 + st_g[i], 0, 0);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if constexpr (IS_FP8) {  // K: raw e4m3; V: e4m3 -> bf16 is exact, 16 elements = two bf16 chunks
        if constexpr (!DMA_K8) lds_write_b128(Kbuf + buf * KTILE + st_k[i], kst[i]);
        lds_write_b128(Vbuf + buf * TILE + st_v[i], fp8x8_to_bf16(u32x2{vst[i][0], vst[i][1]}));
        lds_write_b128(Vbuf + buf * TILE + st_v1[i], fp8x8_to_bf16(u32x2{vst[i][2], vst[i][3]}));
      } else {
        lds_write_b128(Kbuf + buf * KTILE + st_k[i], kst[i]);
        lds_write_b128(Vbuf + buf * TILE + st_v[i], vst[i]);
      }
    }
  };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.0f;
  float m = -INFINITY;     // reference max of this row's scores (may lag the true max by < 2^THR); units: raw scores, PRE: log2 units
  float mthr = -INFINITY;  // m + threshold: a tile max above it forces a rescale
  float l = 0.0f;          // this lane half's share of the running sum
  const float c2 = p.scale * 1.4426950408889634f;  // scale * log2(e)
  const float cm = PRE ? 1.0f : c2;                // m's units -> log2 units
  const float thr_raw = FA_DEFER_THR / cm;         // the threshold in m's units
  // PRE: -m in all 16 registers of a tuple = the C operand of each score chain's first MFMA (0 until the first tile has
  // set m: the first tile's scores come out raw and go through the exact path)
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.0f;
  if constexpr (PRE) asm volatile("" : "+v"(negm));  // opaque: else hipcc re-materialises the splat in front of every MFMA

  if constexpr (DMA) {
    stage_dma(sp, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if constexpr (DMA_K8) stage_dma_k8(sp, 0);
    stage_load(sp);  // this split's first tile (past the end of a short head: zeros through the descriptor, never used)
    stage_write(0);
    if constexpr (DMA_K8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // Retire the Q-fragment loads HERE: hipcc's waitcnt pass otherwise carries them into the
  // loop as "possibly pending" and drains vmcnt(0) in front of every tile's first MFMAs, i.e.
  // waits for the prefetch it has just issued (seen in the .s as vmcnt(3)..vmcnt(0)).
  if constexpr (IS_FP8) {
#pragma unroll
    for (int j = 0; j < NS8; ++j) asm volatile("" : "+v"(qf8[j]));
  } else {
    if constexpr (PRE) {  // Q~ = round(c.Q), once per block
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (elem)((float)qf[ks][j] * c2);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  }
  __syncthreads();

  const float sum_thr = __builtin_exp2f(FA_DEFER_THR);
  // cold path of the softmax: the scores of the current tile once more (plain product, no prefetching)
  auto recompute_scores = [&](auto bufc, f32x16 (&s)[2], const int kv0, const bool need_mask) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.0f;
      if constexpr (IS_FP8) {
#pragma unroll
        for (int j = 0; j < NS8; ++j) {
          const lds_char *kr = Kbuf + buf * KTILE + (32 * kb + r) * KRB;
          const u32x4 a = lds_read_b128(kr + (((4 * j + 2 * h) ^ kx8) << 4));
          const u32x4 b = lds_read_b128(kr + (((4 * j + 2 * h + 1) ^ kx8) << 4));
          const i32x8 kf8 = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
          s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf8, qf8[j], s[kb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const vec8 kf = __builtin_bit_cast(vec8, lds_read_b128(kptr[ks] + buf * KTILE + kb * 32 * RB));
          s[kb] = M::mfma(kf, qf[ks], s[kb]);
        }
      }
    }
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        int lim = p.Nk - 1 - kv0 - 32 * kb - 4 * h;
        if (CAUSAL) lim = min(lim, qrow + coff - kv0 - 32 * kb - 4 * h);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kpart = (i & 3) + 8 * (i >> 2);
          s[kb][i] = (kpart > lim) ? -INFINITY : s[kb][i];
        }
      }
    }
  };
  // One KV tile; BUF (the LDS buffer holding tile t) is a compile-time constant so every
  // LDS address is a per-lane base register plus an immediate.
  auto tile = [&](auto bufc, const int t) {
    constexpr int buf = decltype(bufc)::value;
    const int kv0 = t * BN;
    if (t + SPLIT < nT) {  // this split's next tile, in flight under this tile's MFMAs (issuing the LDS-DMA pieces behind the score
      if constexpr (DMA) {  // MFMAs instead was measured: -29 % on causal shapes, profiles/r03/ab_dma_late_and_d128_auto.log)
        stage_dma(t + SPLIT, buf ^ 1);
      } else {
        if constexpr (DMA_K8) stage_dma_k8(t + SPLIT, buf ^ 1);
        stage_load(t + SPLIT);
      }
    }

    // whole-tile skip per wave (kernels.metal:682 with Br = 32): every key of
    // the tile is past this wave's last query row
    const bool wave_active = (SPLIT == 1 || t < nT) && (!CAUSAL || (kv0 <= qw0 + WM - 1 + coff));
    if (wave_active) {
      const lds_char *Kt = Kbuf + buf * KTILE;  // (fp8 score path only)
      (void)Kt;
      // ---- S^T = K.Q^T : s[kb][reg] = S[q = r][key = kv0 + 32kb + (reg&3) + 8(reg>>2) + 4h]
      // All K fragment reads are issued before the first MFMA, and (D = 64) the V^T
      // fragments of the PV product are streamed in between the MFMAs, two transposed
      // reads per MFMA: they do not depend on the softmax, so PV finds its operands in
      // registers instead of waiting on LDS per MFMA. sched_barrier pins that order.
#if FA_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      f32x16 s[2];
      s16x4 vlo[VPRE ? 2 : 1][2][DB], vhi[VPRE ? 2 : 1][2][DB];
      if constexpr (VPRE) {
        vec8 kf[2][KS];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            kf[kb][ks] = __builtin_bit_cast(vec8, lds_read_b128(kptr[ks] + buf * KTILE + kb * 32 * RB));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) s[kb][i] = 0.0f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s[kb] = M::mfma(kf[kb][ks], qf[ks], (PRE && ks == 0) ? negm : s[kb]);
            if constexpr (FA_VPRE_HALF) {  // the 2 * DB V blocks of keys 0..31, one per two QK MFMAs
              const int mm = kb * KS + ks;
              if ((mm & 1) == 0) {
                const int m = mm >> 1;
                const int vst = (m / DB) % 2, vdb = m % DB;
                const lds_char *vb = vptr[vdb] + buf * TILE + (16 * vst) * RB;
                vlo[0][vst][vdb] = lds_read_tr16(vb);
                vhi[0][vst][vdb] = lds_read_tr16(vb + 8 * RB);
              }
            } else {
              constexpr int PER = (2 * 2 * DB) / (2 * KS);  // V block reads per QK MFMA (1 at D=64)
#pragma unroll
              for (int u = 0; u < PER; ++u) {
                const int m = (kb * KS + ks) * PER + u;
                const int vkb = m / (2 * DB), vst = (m / DB) % 2, vdb = m % DB;
                const lds_char *vb = vptr[vdb] + buf * TILE + (32 * vkb + 16 * vst) * RB;
                vlo[vkb][vst][vdb] = lds_read_tr16(vb);
                vhi[vkb][vst][vdb] = lds_read_tr16(vb + 8 * RB);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if constexpr (IS_FP8) {
        // e4m3 score product: one scaled MFMA per 32 x 32 block and 64 head-dim elements (unit scales: 2^0)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) s[kb][i] = 0.0f;
#pragma unroll
          for (int j = 0; j < NS8; ++j) {
            const lds_char *kr = Kt + (32 * kb + r) * KRB;
            const u32x4 a = lds_read_b128(kr + (((4 * j + 2 * h) ^ kx8) << 4));
            const u32x4 b = lds_read_b128(kr + (((4 * j + 2 * h + 1) ^ kx8) << 4));
            const i32x8 kf8 = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
            s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf8, qf8[j], s[kb], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          }
        }
      } else {
        // head_dim 128: 16 K fragments would cost 64 VGPRs if held live; read each one LA MFMAs ahead of
        // its use instead (profile before: 9 % of wave time stalled on LDS issue, 14 % MFMA/VALU co-execution)
        constexpr int NK = 2 * KS, LA = FA_LAK;
        vec8 kf[NK];
        auto kread = [&](int i) { kf[i] = __builtin_bit_cast(vec8, lds_read_b128(kptr[i % KS] + buf * KTILE + (i / KS) * 32 * RB)); };
#pragma unroll
        for (int i = 0; i < LA; ++i) kread(i);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          s[i / KS] = M::mfma(kf[i], qf[i % KS], (i % KS) == 0 ? (PRE ? negm : zero) : s[i / KS]);
          if (i + LA < NK) kread(i + LA);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- mask (only on tiles that cross the diagonal or the end of the sequence)
      const bool need_mask = (CAUSAL && (kv0 + BN - 1 > qw0 + coff)) || (kv0 + BN > p.Nk);
      if (need_mask) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          // masked iff key > qrow (kernels.metal:748) or key >= N. A 32 x 32 block whose every key is visible to every
          // row of the wave needs no work (wave-uniform test: of the two blocks of a diagonal tile one is of this
          // kind or entirely masked): key offsets inside the block span 0..31, the wave's rows qw0..qw0+31.
          const int xw = min(p.Nk - 1 - kv0 - 32 * kb, CAUSAL ? qw0 + coff - kv0 - 32 * kb : 0x7fffffff);
          if (xw >= 31) continue;
          int lim = p.Nk - 1 - kv0 - 32 * kb - 4 * h;
          if (CAUSAL) lim = min(lim, qrow + coff - kv0 - 32 * kb - 4 * h);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int kpart = (i & 3) + 8 * (i >> 2);
            s[kb][i] = (kpart > lim) ? -INFINITY : s[kb][i];
          }
        }
      }
#if FA_PRIO == 1
      __builtin_amdgcn_s_setprio(1);
#elif FA_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      // ---- online softmax, lane-local + one half swap
      const float mc_old = m * c2;  // (unused with PRE)
      (void)mc_old;
      float ls0 = 0.0f, ls1 = 0.0f;
#if FA_SUMTRIG
      // Deferred row max (T13) WITHOUT a per-tile max: P = exp2(c.s - c.m) is formed with the running reference m, and
      // the row sums that are needed anyway tell whether m is stale: a lane whose 32 probabilities add up to more than
      // 2^THR (or to +inf) has a score more than 2^THR above m at worst. Only then -- and on the first tile, where m is
      // -inf -- the wave takes the exact path: recompute the scores (they were overwritten by P; K is still in LDS),
      // take the row max, rescale O and l, form P again. Saves the 16 v_max3 + swap + compare of every tile
      // (141 -> 118 VALU per 16 MFMAs at head_dim 64); m, l and O stay mutually consistent, LSE = m.scale + ln(l) is exact.
      bool exact = (t == sp);  // this split's first tile
      if (t != sp) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if constexpr (PRE) {  // the matrix core has already subtracted m
            s[0][i] = __builtin_amdgcn_exp2f(s[0][i]);
            s[1][i] = __builtin_amdgcn_exp2f(s[1][i]);
          } else {
            s[0][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][i], c2, -mc_old));
            s[1][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[1][i], c2, -mc_old));
          }
          ls0 = (i == 0) ? s[0][i] : ls0 + s[0][i];  // (0 + x is an instruction: -ffp-contract/-0.0 rules keep it)
          ls1 = (i == 0) ? s[1][i] : ls1 + s[1][i];
        }
        exact = __builtin_amdgcn_ballot_w64(ls0 + ls1 > sum_thr) != 0;  // wave-uniform
        if (exact) recompute_scores(bufc, s, kv0, need_mask);
      }
      if (exact) {
#else
      {
#endif
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);  // -> v_max3_f32
        {
          float lo, hi;
          half_pair(mx, lo, hi);
          mx = fmaxf(lo, hi);
        }
        // deferred row max (T13): O and l are rescaled only when some row's tile max exceeds the running
        // reference m by more than 2^THR (log2 domain); otherwise p = exp2(c.s - c.m) <= 2^THR with the
        // stale m. m, l and O stay mutually consistent, so LSE = m.scale + ln(l) is exact either way.
        // On random data a 32-row wave sees SOME row's max move in most tiles, so the exact form
        // (rescale whenever a max moved) paid the 32-multiply O pass nearly every tile.
        if (FA_SUMTRIG || __builtin_amdgcn_ballot_w64(mx > mthr) != 0) {  // wave-uniform; first tile: mthr = -inf
          const float m_new = fmaxf(m, mx);
          const float alpha = __builtin_amdgcn_exp2f((m - m_new) * cm);
          l *= alpha;
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
          m = m_new;
          mthr = m_new + thr_raw;
          if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[i] = -m_new;
            asm volatile("" : "+v"(negm));
          }
        }
        const float mc = m * cm;
        ls0 = 0.0f;
        ls1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[0][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][i], cm, -mc));
          s[1][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[1][i], cm, -mc));
          ls0 += s[0][i];
          ls1 += s[1][i];
        }
      }
      l += ls0 + ls1;
#if FA_PRIO == 1
      __builtin_amdgcn_s_setprio(0);
#elif FA_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      // ---- O^T += V^T.P^T : per 16-key step, P fragment = 8 accumulator registers
      if constexpr (VPRE) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            vec8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (elem)s[kb][8 * st + j];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
              const s16x8 v8 = __builtin_shufflevector(vlo[kb][st][db], vhi[kb][st][db], 0, 1, 2, 3, 4, 5, 6, 7);
              oacc[db] = M::mfma(__builtin_bit_cast(vec8, v8), pf, oacc[db]);
              if constexpr (FA_VPRE_HALF) {
                if (kb == 0) {  // keys 32..63: the same block of the second half, read behind the MFMA that frees its registers' twin
                  const lds_char *vb = vptr[db] + buf * TILE + (32 + 16 * st) * RB;
                  vlo[1][st][db] = lds_read_tr16(vb);
                  vhi[1][st][db] = lds_read_tr16(vb + 8 * RB);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          }
        }
      } else {
        // V^T fragments just in time, LA MFMAs ahead (step j = (kb, st, db))
        constexpr int NV = 2 * 2 * DB, LA = FA_LAV;
        s16x4 wlo[NV], whi[NV];
        auto vread = [&](int j) {
          const lds_char *vb = vptr[j % DB] + buf * TILE + (32 * (j / (2 * DB)) + 16 * ((j / DB) % 2)) * RB;
          wlo[j] = lds_read_tr16(vb);           // keys +4h+0..3   (k elements 0..3)
          whi[j] = lds_read_tr16(vb + 8 * RB);  // keys +8+4h+0..3 (k elements 4..7)
        };
        vec8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[kb][st][j] = (elem)s[kb][8 * st + j];
#pragma unroll
        for (int j = 0; j < LA; ++j) vread(j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const s16x8 v8 = __builtin_shufflevector(wlo[j], whi[j], 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[j % DB] = M::mfma(__builtin_bit_cast(vec8, v8), pf[j / (2 * DB)][(j / DB) % 2], oacc[j % DB]);
          if (j + LA < NV) vread(j + LA);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (t + SPLIT < nT) {
      if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces issued at the top of this tile have landed
      } else {
        stage_write(buf ^ 1);
        if constexpr (DMA_K8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  };
  const int steps = (nT + SPLIT - 1) / SPLIT;  // every wave runs the same number of steps (one barrier each)
  for (int j = 0; j < steps; j += 2) {
    tile(std::integral_constant<int, 0>{}, j * SPLIT + sp);
    if (j + 1 < steps) tile(std::integral_constant<int, 1>{}, (j + 1) * SPLIT + sp);
  }

  if constexpr (SPLIT == 2) {
    // ---- merge the two splits: waves 4-7 publish (O^T, m, l) lane by lane, waves 0-3 fold them into their own by the
    // reference maxima (a split that saw no tile has m = -inf, l = 0: weight 0). The buffer sits behind the epilogue's
    // O tiles; the K/V buffers are free (last step's barrier).
    constexpr int NACC = 16 * DB;
    float *mb = (float *)((lds_char *)smem_generic + RW * WM * RB) + (size_t)wave * (NACC + 2) * 64 + lane;
    if (sp == 1) {
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) mb[(16 * db + i) * 64] = oacc[db][i];
      mb[NACC * 64] = m;
      mb[(NACC + 1) * 64] = l;
    }
    __syncthreads();
    if (sp == 0) {
      const float m1 = mb[NACC * 64], l1 = mb[(NACC + 1) * 64];
      const float mm = fmaxf(m, m1);  // split 0 owns tile 0: m is finite
      const float a0 = __builtin_amdgcn_exp2f((m - mm) * cm), a1 = __builtin_amdgcn_exp2f((m1 - mm) * cm);
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = oacc[db][i] * a0 + mb[(16 * db + i) * 64] * a1;
      l = l * a0 + l1 * a1;
      m = mm;
    }
  }

  // ---- epilogue: normalise, LSE, O tile -> LDS -> coalesced 16-byte stores (split2: waves 0-3 only; the others keep
  // the barrier company)
  lds_char *Ot = (lds_char *)smem_generic + wave * (WM * RB);  // this wave's [32][D] tile (inside the K buffers)
  if (sp == 0) {
    {
      float lo, hi;
      half_pair(l, lo, hi);
      l = lo + hi;
    }
    const float inv_l = 1.0f / l;
    if (p.lse != nullptr && h == 0 && qrow < p.N)
      p.lse[(long long)bh * p.N + qrow] = m * (PRE ? 0.6931471805599453f : p.scale) + logf(l);
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        // registers 4g4..4g4+3 = d columns 32db + 8g4 + 4h + 0..3 of row r
        elem e0 = (elem)(oacc[db][4 * g4 + 0] * inv_l), e1 = (elem)(oacc[db][4 * g4 + 1] * inv_l);
        elem e2 = (elem)(oacc[db][4 * g4 + 2] * inv_l), e3 = (elem)(oacc[db][4 * g4 + 3] * inv_l);
        u32x2 w;
        w[0] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
        w[1] = (unsigned)__builtin_bit_cast(unsigned short, e2) | ((unsigned)__builtin_bit_cast(unsigned short, e3) << 16);
        // 16-byte chunk index XOR (r & (CPR-1)) spreads the rows over the banks
        const int col_b = (32 * db + 8 * g4 + 4 * h) * 2;
        const int ch = (col_b >> 4) ^ (r & (CPRL - 1));
        lds_write_b64(Ot + r * RB + (ch << 4) + (col_b & 15), w);
      }
    }
  }
  __syncthreads();
  if (sp == 0) {
    elem *Og = (elem *)p.o + base;
#pragma unroll
    for (int it = 0; it < WM * CPR / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR, ch = idx % CPR;
      const u32x4 vv = lds_read_b128(Ot + row * RB + ((ch ^ (row & (CPRL - 1))) << 4));
      if (qw0 + row < p.N)
        *reinterpret_cast<u32x4 *>(Og + (long long)(qw0 + row) * D + ch * 8) = vv;
    }
  }
}

template <typename Tag, int D, bool CAUSAL, bool PRESC>
__global__ __launch_bounds__(NTHREADS, (main_kernel_occ4<Tag, D, PRESC>() ? 4 : D <= 64 ? 3 : D <= 128 ? 2 : 1)) void fwd_mfma_kernel(Params p) {
  fwd_mfma_body<Tag, D, CAUSAL, 1, PRESC>(p);
}

template <typename Tag, int D, bool CAUSAL>
__global__ __launch_bounds__(2 * NTHREADS, 2) void fwd_mfma_split2_kernel(Params p) {
  fwd_mfma_body<Tag, D, CAUSAL, 2, true>(p);
}

template <typename Tag, int D, bool CAUSAL>
__global__ __launch_bounds__(NTHREADS, 2) void fwd_mfma_h64s2_kernel(Params p) {
  fwd_mfma_body<Tag, D, CAUSAL, 2, true, 64>(p);
}

// ---------------------------------------------------------------------------
bool mfma_supported(int dtype, int D) {
  if (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) return D == 32 || D == 64 || D == 96 || D == 128 || D == 256;
  return dtype == FA_DTYPE_FP8_E4M3 && (D == 64 || D == 128 || D == 256);  // an fp8 row must fill whole 16-byte chunks per thread
}

template <typename Tag, int D, bool CAUSAL, bool PRESC>
static hipError_t launch_one_(const Params &p, hipStream_t s) {
  const int nQ = (p.N + BM - 1) / BM;
  const size_t vrow = (D == 96) ? 256 : D * 2;  // fp8: K tiles stay e4m3 (rows of D bytes), V tiles are widened to bf16
  const size_t smem = 2 * BN * (std::is_same<Tag, FP8>::value ? (size_t)D : vrow) + 2 * BN * vrow;
  auto kern = fwd_mfma_kernel<Tag, D, CAUSAL, PRESC>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem);
    if (e != hipSuccess) return e;
  }
  Params pp = p;
  pp.head_group = causal_head_group(p, D, std::is_same<Tag, FP8>::value ? 1 : 2);
#ifdef FA_DEBUG_KNOBS  // scheduling experiments only: never compiled into the shipped library
  static const int env_head_group = [] { const char *e = getenv("FA_HEAD_GROUP"); return e ? atoi(e) : -1; }();
  if (env_head_group >= 0) pp.head_group = env_head_group;
#ifdef FA_FORCE_HEAD_GROUP  // compile-time form for tools/ab.py (several builds side by side in one process share the environment)
  pp.head_group = FA_FORCE_HEAD_GROUP;
#endif
#endif
  set_block_divisors(pp, nQ, pp.head_group);
  (void)hipGetLastError();  // do not report an older sticky error as this launch's
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, pp);
  return hipGetLastError();
}

// p.exact (variant mfma_exact) selects the instantiation without the pre-scaled operand; where the pre-scaling does not
// exist (fp8, head_dim 256) there is only that one.
template <typename Tag, int D, bool CAUSAL>
static hipError_t launch_one(const Params &p, hipStream_t s) {
  if constexpr (prescale_applies<Tag, D>()) {
    if (!p.exact) return launch_one_<Tag, D, CAUSAL, true>(p, s);
  }
  return launch_one_<Tag, D, CAUSAL, false>(p, s);
}

template <typename Tag, int D, bool CAUSAL>
static hipError_t launch_split2_one(const Params &p, hipStream_t s) {
  const int nQ = (p.N + BM - 1) / BM;
  const size_t vrow = D * 2;
  const size_t group = 2 * BN * (std::is_same<Tag, FP8>::value ? (size_t)D : vrow) + 2 * BN * vrow;  // as launch_one
  const size_t merge_end = (size_t)(BM / WM) * WM * vrow + (size_t)(BM / WM) * (16 * (D / 32) + 2) * 64 * 4;
  const size_t smem = std::max(2 * group, merge_end);
  auto kern = fwd_mfma_split2_kernel<Tag, D, CAUSAL>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem);
    if (e != hipSuccess) return e;
  }
  Params pp = p;
  pp.head_group = 0;
  set_block_divisors(pp, nQ, 0);
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(2 * NTHREADS), smem, s, pp);
  return hipGetLastError();
}

template <typename Tag, int D, bool CAUSAL>
static hipError_t launch_h64s2_one(const Params &p, hipStream_t s) {
  constexpr int ROWS = 64;
  const int nQ = (p.N + ROWS - 1) / ROWS;
  const size_t vrow = D * 2;
  const size_t group = 2 * BN * (std::is_same<Tag, FP8>::value ? (size_t)D : vrow) + 2 * BN * vrow;  // as launch_one
  const size_t merge_end = (size_t)ROWS * vrow + (size_t)(ROWS / WM) * (16 * (D / 32) + 2) * 64 * 4;
  const size_t smem = std::max(2 * group, merge_end);
  auto kern = fwd_mfma_h64s2_kernel<Tag, D, CAUSAL>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, (int)smem);
    if (e != hipSuccess) return e;
  }
  Params pp = p;
  pp.head_group = 0;
  set_block_divisors(pp, nQ, 0);
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, pp);
  return hipGetLastError();
}

bool mfma_h64s2_supported(int dtype, int D) { return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) && D == 64; }

hipError_t launch_mfma_h64s2(const Params &p, int dtype, hipStream_t s) {
  if (p.D != 64) return hipErrorInvalidValue;
  if (dtype == FA_DTYPE_F16) return p.is_causal ? launch_h64s2_one<F16, 64, true>(p, s) : launch_h64s2_one<F16, 64, false>(p, s);
  return p.is_causal ? launch_h64s2_one<BF16, 64, true>(p, s) : launch_h64s2_one<BF16, 64, false>(p, s);
}

bool mfma_split2_supported(int dtype, int D) {
  return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16 || dtype == FA_DTYPE_FP8_E4M3) && (D == 64 || D == 128);
}

hipError_t launch_mfma_split2(const Params &p, int dtype, hipStream_t s) {
  auto go = [&](auto tag) -> hipError_t {
    using Tag = decltype(tag);
    if (p.D == 64) return p.is_causal ? launch_split2_one<Tag, 64, true>(p, s) : launch_split2_one<Tag, 64, false>(p, s);
    if (p.D == 128) return p.is_causal ? launch_split2_one<Tag, 128, true>(p, s) : launch_split2_one<Tag, 128, false>(p, s);
    return hipErrorInvalidValue;
  };
  if (dtype == FA_DTYPE_FP8_E4M3) return go(FP8{});
  return dtype == FA_DTYPE_F16 ? go(F16{}) : go(BF16{});
}

template <typename Tag>
static hipError_t launch_dt(const Params &p, hipStream_t s) {
  constexpr bool IS8 = std::is_same<Tag, FP8>::value;
  switch (p.D) {
    case 64: return p.is_causal ? launch_one<Tag, 64, true>(p, s) : launch_one<Tag, 64, false>(p, s);
    case 128: return p.is_causal ? launch_one<Tag, 128, true>(p, s) : launch_one<Tag, 128, false>(p, s);
    case 256: return p.is_causal ? launch_one<Tag, 256, true>(p, s) : launch_one<Tag, 256, false>(p, s);
    case 32: if constexpr (!IS8) return p.is_causal ? launch_one<Tag, 32, true>(p, s) : launch_one<Tag, 32, false>(p, s); break;
    case 96: if constexpr (!IS8) return p.is_causal ? launch_one<Tag, 96, true>(p, s) : launch_one<Tag, 96, false>(p, s); break;
  }
  return hipErrorInvalidValue;
}

hipError_t launch_mfma(const Params &p, int dtype, hipStream_t s) {
  if (dtype == FA_DTYPE_FP8_E4M3) return launch_dt<FP8>(p, s);
  return dtype == FA_DTYPE_F16 ? launch_dt<F16>(p, s) : launch_dt<BF16>(p, s);
}

}  // namespace fa
