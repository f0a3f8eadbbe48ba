// fa_bwd_kernels.hip -- backward of the operator on the CDNA4 matrix cores (head_dim 64 and 128).
//
// Replaces /root/reference/kernels.metal:905-1265 (flash_attention_backward_kernel): same
// math -- D_i = rowsum(dO o O) (:983-990), P = exp(S*scale - L_i) recomputed from the forward's
// LSE (:1082-1089), dV += P^T dO, dP = dO V^T, dS = P o (dP - D_i) * scale (:1160-1169),
// dQ += dS K, dK += dS^T Q -- nothing else. The reference flushes dK/dV with global float
// atomics from every Q block (:1221-1246); here no gradient is accumulated across workgroups:
//
//   bwd_dq_kernel      one workgroup per 128 query rows, loops over KV tiles (like the forward):
//                        delta[b,h,i] = sum_d dO*O of its rows, formed in the prologue from the dO fragments it holds
//                        anyway and left in the workspace (B*H*N floats) for the dK/dV kernel (round 2: a kernel of its own),
//                        S^T = K.Q^T, dP^T = V.dO^T, dS^T = P^T o (dP^T - delta) * scale,
//                        dQ^T += K^T.dS^T
//   bwd_dkdv_kernel    one workgroup per 128 keys, loops over Q tiles:
//                        S = Q.K^T, dP = dO.V^T, P, dS (as above),
//                        dV^T += dO^T.P,  dK^T += Q^T.dS
//
// S and dP are computed twice (7 matrix products instead of 5) in exchange for deterministic,
// atomic-free, bitwise reproducible gradients. (The single-pass 5-product form sums dQ across key blocks with float
// atomics: at head_dim 64 that is 570 MB of adds for BASELINE config 3's shape with 256 keys per workgroup, 436 us MEASURED
// (tools/probes/probe_dq_atomic_floor.hip: 1.31-1.34 TB/s, MI355X_MICROARCH.md, Global float atomics) -- as long as both kernels here
// together; 861 us vs ~850 at head_dim 128: the bytes per FLOP do not depend on the head dim. DESIGN 4.6b.)
// Round 3: the row constants ride in the accumulators (cdna guide, attention backward): the operand held in registers is
// pre-scaled by scale*log2(e) (Q~ in the dQ kernel -- bit for bit the forward's operand --, K~ in the dK/dV kernel) and the
// score chains start from -LSE*log2(e), the dP chains from -delta, so P = exp2(S') and dS = P * dP' are one
// transcendental and one multiply per score; the softmax scale is applied once to the finished dQ / dK. Both kernels reuse the forward's machinery
// (fa_mfma_kernel.hip): one operand's fragments live in registers, the other side streams
// through double-buffered LDS tiles; the score tile comes out of v_mfma_f32_32x32x16 with the
// reduction index of the NEXT product in its registers, so P / dS feed that product as the B
// operand without leaving the register file, and the transposed A operands (K^T, dO^T, Q^T) are
// ds_read_b64_tr_b16 reads of row-major tiles. A tile that is read both by rows and transposed sits in LDS ONCE, under a
// chunk swizzle that keeps both kinds of read (and the staging stores) free of bank conflicts (cdna guide T10, "one image
// for row reads and transposed reads"; round 2 kept two images and paid the LDS store bandwidth twice -- these kernels
// are LDS-bound: every MFMA takes a 1 KiB fragment from LDS, half the LDS bandwidth at full matrix rate, before any store).
// Second half of round 3: each 64-row sub-tile is worked one 32-row half at a time (scores, P / dS, then that half's share
// of the second products), so one score and one dP tuple are live; the per-lane LDS addresses are ABSOLUTE addresses in the
// current buffer, moved to the other buffer in place once per tile. Together: 160-166 registers, three workgroups per CU at
// head_dim 64 (config-3 shape 692 -> 811 TFLOP/s with the SLP vectorizer off, profiles/r03/ab_bwd_*.log).
// Grouped-query heads (fa_bwd_ex): the dK/dV workgroup of a key/value head visits its H / Hkv query heads in turn.
#include "fa_mfma_common.h"

namespace fa {

struct BwdParams {
  const void *q, *k, *v, *o, *d_o;
  const float *lse;
  float *dq, *dk, *dv;
  float *delta;  // workspace [B,H,N]
  int B, H, N, D;  // H = query heads, N = query rows per head
  int Nk;          // keys per head (causal: bottom-right aligned, key j visible to query i iff j <= i + Nk - N; Nk >= N then)
  float scale;
  long long batch_stride, head_stride;  // of Q, O, dO, dQ (elements)
  int is_causal;
  int Hkv;                                    // key/value heads: query head h reads (and dK/dV sum over) key head h / (H / Hkv)
  long long kv_batch_stride, kv_head_stride;  // of K, V, dK, dV
};

constexpr float LOG2E = 1.4426950408889634f;

#ifndef FA_BWD_DMA
#define FA_BWD_DMA 1  // 1: the streamed tiles go global -> LDS by LDS-DMA (buffer_load ... lds; the chunk swizzle sits on the source address):
#endif                // no staging registers, no ds_write_b128 (as in the forward kernels, profiles/r03/ab_mfma_lds_dma.log); 0 = register staging
#ifndef FA_BWD_LA
#define FA_BWD_LA 3  // row fragments are read this many MFMAs ahead of their use
#endif
#ifndef FA_BWD_LA2
#define FA_BWD_LA2 2  // the same for the transposed fragments
#endif
#ifndef FA_BWD_KV128_LA
#define FA_BWD_KV128_LA 1  // the head_dim-128 dK/dV kernel (256 registers at two workgroups per CU) affords one step of each
#define FA_BWD_KV128_LA2 1
#endif
// Workgroups per CU the kernels are compiled for: three at head_dim 64 (register cap 168, 32-34 KiB of LDS), two at 128
// (cap 256, 64-66 KiB). Round 3 first ran two / one (205-214 registers): see bwd_dq_kernel for what brought them down.
constexpr int bwd_occ(int D) { return D == 64 ? 3 : D == 128 ? 2 : 1; }  // (head_dim 256: one workgroup per CU, 512 registers, 128 KiB of LDS)
// 64-row sub-tiles per staged tile: one barrier and one staging pass per SUB * 64 keys (dQ) / queries (dK, dV). 2 paid while
// one workgroup per CU exposed every barrier (+16 % at head_dim 128 then); at the occupancy above 1 is faster and is what
// fits the LDS (profiles/r03/ab_bwd_dq_per_half.log).
#ifndef FA_BWD_SUB
#define FA_BWD_SUB 1
#endif
constexpr int bwd_sub_dq(int D) { return FA_BWD_SUB; }
constexpr int bwd_sub_kv(int D) { return FA_BWD_SUB; }

// per-head-dim constants of the kernels below (the reference kernel is head_dim 64 only, kernels.metal:905-1265;
// 128 is the same algorithm with twice the k-steps / output blocks and two workgroups per CU)
#define FA_BWD_CONSTS(D, SUBS)                                                                   \
  constexpr int BSUB = (SUBS);     /* sub-tiles per staged tile */                               \
  constexpr int BT = BSUB * BN;    /* rows of a staged tile */                                   \
  constexpr int BD = (D);          /* head dim */                                                \
  constexpr int BRB = BD * 2;      /* row bytes */                                               \
  constexpr int BCPR = BD / 8;     /* 16-byte chunks per row */                                  \
  constexpr int BKS = BD / 16;     /* k-steps over the head dim */                               \
  constexpr int BDB = BD / 32;     /* 32-wide output blocks over the head dim */                 \
  constexpr int BTILE = BN * BRB;  /* one 64-row sub-tile image */                               \
  constexpr int STILE = BSUB * BTILE; /* one staged tile (BT rows) */                             \
  /* XOR on the 16-byte chunk index of a row: conflict-free for ds_read_b128 row reads, ds_read_b64_tr_b16 and ds_write_b128 */ \
  auto u_swz = [](int row) { return BD == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 3) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3)); }; \
  /* transposed read of the 4-row x 32-column block (R0 + 4h + vq, columns 32db ..), R0 a multiple of 8: the swizzle's low   */ \
  /* bits depend on R0 only through `variant` = (R0 >> 3) & 3 (head_dim 64) or (R0 >> 3) & 1 (head_dim 128): NTV base addresses */ \
  constexpr int NTV = BD == 64 ? 4 : 2;                                                       \
  auto tr_off = [&](int variant, int db, int h_, int g1_, int vq_, int vp_) {                 \
    const int row = 8 * variant + 4 * h_ + vq_; /* a representative R0 = 8 * variant */        \
    return (4 * h_ + vq_) * BRB + ((((4 * db) + 2 * g1_ + (vp_ >> 1)) ^ u_swz(row)) << 4) + 8 * (vp_ & 1); \
  };                                                                                          \
  /* LDS-DMA: wave w moves the 1-KiB pieces w, w+4, ... of a staged tile; lane L fills LDS bytes [16 L, 16 L + 16) of its    */ \
  /* piece = row RPP w + L / BCPR, physical chunk L % BCPR, which holds logical chunk (L % BCPR) ^ u_swz(row); the swizzle */ \
  /* does not depend on the piece index (4 RPP rows per step of the piece index: a multiple of its period)                */ \
  constexpr int RPP = 1024 / BRB, NPW = (BT / RPP) / 4;                                       \
  auto dma_off = [&](int wave_, int lane_) {                                                  \
    const int row = wave_ * RPP + lane_ / BCPR, pc = lane_ % BCPR;                            \
    return (unsigned)(row * BRB + ((pc ^ u_swz(row)) << 4));                                  \
  };                                                                                          \
  (void)BT; (void)BCPR; (void)BKS; (void)BDB; (void)BTILE; (void)STILE; (void)u_swz; (void)tr_off; (void)NTV; (void)RPP; (void)NPW; (void)dma_off

// Head dims other than 64 / 128 (any multiple of 8 up to 128: 32, 96, ...) run the next larger instantiation on ZERO-PADDED rows
// (PAD): rows keep their packed pitch of p.D elements in global memory; the LDS images and the register fragments have the
// kernel's pitch, and every 16-byte chunk at or past column p.D is fetched from an offset outside the buffer descriptor's range,
// which reads as zeros (register fragments and LDS-DMA alike). The padding columns then add 0 to every score and receive
// gradients that are never stored. Same arithmetic per real column, (BD - D) / BD of the matrix work wasted (D = 96: a quarter).
#define FA_BWD_PAD(PAD_)                                                                                      \
  const int GRB = (PAD_) ? p.D * 2 : BRB; /* row bytes in global memory */                                    \
  constexpr unsigned OOB = 0x80000000u;   /* past any head (bwd_impl keeps padded heads below 2 GiB) */       \
  auto gcol = [&](int chunk) -> unsigned { return (!(PAD_) || chunk * 8 < p.D) ? (unsigned)chunk * 16 : OOB; }; \
  auto dma_off_pad = [&](int wave_, int lane_) {                                                              \
    const int row = wave_ * RPP + lane_ / BCPR, lc = (lane_ % BCPR) ^ u_swz(row);                             \
    return (unsigned)(row * GRB) + gcol(lc);                                                                  \
  };                                                                                                          \
  (void)gcol; (void)dma_off_pad; (void)OOB

// ---------------------------------------------------------------------------
// dQ: workgroup = 128 query rows, wave = 32 rows (query on the lane, keys in the registers)
// ---------------------------------------------------------------------------
template <typename Tag, int D, bool CAUSAL, bool PAD>
__global__ __launch_bounds__(NTHREADS, bwd_occ(D)) void bwd_dq_kernel(BwdParams p) {
  FA_BWD_CONSTS(D, bwd_sub_dq(D));
  FA_BWD_PAD(PAD);
  using M = MT<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  lds_char *KU = smem;               // [2] K tile of BT rows (read by rows for S, transposed for dQ)
  lds_char *VR = smem + 2 * STILE;   // [2] V tile (read by rows)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nQ = (p.N + BM - 1) / BM;
  int bh, qb;
  map_block_div<CAUSAL>(blockIdx.x, p.B * p.H, nQ, bh, qb);
  const long long base = (long long)(bh / p.H) * p.batch_stride + (long long)(bh % p.H) * p.head_stride;
  const long long base_kv = (long long)(bh / p.H) * p.kv_batch_stride + (long long)((bh % p.H) / (p.H / p.Hkv)) * p.kv_head_stride;
  const int q0 = qb * BM, qw0 = q0 + wave * WM, qrow = qw0 + r;

  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const int coff = p.Nk - p.N;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.q + base), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.k + base_kv), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.v + base_kv), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.d_o + base), 0, head_bytes, 0x00020000);

  vec8 qf[BKS], dof[BKS];  // B operands: lane (r,h) holds row qrow, columns 16ks+8h..
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) {
    qf[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * GRB + gcol(2 * ks + h), 0, 0));
    dof[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rdo, (unsigned)qrow * GRB + gcol(2 * ks + h), 0, 0));
  }
  // S' = K.Q~ - lse*log2e straight out of the matrix core (rows past N: -inf, p = 0); dP' = V.dO - delta likewise
  const bool qvalid = qrow < p.N;
  const float lse2 = qvalid ? p.lse[(long long)bh * p.N + qrow] * LOG2E : INFINITY;
  const float c2 = p.scale * LOG2E;
  // delta_i = rowsum(dO o O) (kernels.metal:983-990): this lane holds half of row i's dO (columns 16ks + 8h ..), loads the
  // same half of O, and the two halves of the row meet through one permlane swap; written once for the dK/dV kernel
  float dlt = 0.0f;
  {
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.o + base), 0, head_bytes, 0x00020000);
#pragma unroll
    for (int ks = 0; ks < BKS; ++ks) {
      const vec8 of = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(ro, (unsigned)qrow * GRB + gcol(2 * ks + h), 0, 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) dlt = __builtin_fmaf((float)of[j], (float)dof[ks][j], dlt);
    }
    float lo, hi;
    half_pair(dlt, lo, hi);
    dlt = lo + hi;
    if (qvalid && h == 0) p.delta[(long long)bh * p.N + qrow] = dlt;
  }
  f32x16 nlse, ndlt;  // the row constants, one per lane, in all 16 registers of a tuple: C operands of the chains' first MFMAs
#pragma unroll
  for (int i = 0; i < 16; ++i) { nlse[i] = -lse2; ndlt[i] = -dlt; }
  asm volatile("" : "+v"(nlse), "+v"(ndlt));  // opaque: else hipcc re-materialises the splats in front of every MFMA

  const int kx = u_swz(r);
  // ABSOLUTE LDS addresses in the current K buffer (the V image is 2 STILE further), flipped in place once per tile: with the
  // buffer base added at the point of use hipcc kept a second, per-tile copy of all (base + offset) in registers (seen in the ISA)
  const unsigned ku0 = (unsigned)(__UINTPTR_TYPE__)KU;
  int flip = STILE;
  auto at = [](unsigned a) { return (const lds_char *)(__UINTPTR_TYPE__)a; };
  unsigned koff[BKS];
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) koff[ks] = ku0 + r * BRB + (((2 * ks + h) ^ kx) << 4);
  const int g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  unsigned voff[NTV][BDB];
#pragma unroll
  for (int tv = 0; tv < NTV; ++tv)
#pragma unroll
    for (int db = 0; db < BDB; ++db) voff[tv][db] = ku0 + tr_off(tv, db, h, g1, vq, vp);
  constexpr int NCH = BT * BCPR / NTHREADS;
  int st_g[NCH], st_r[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHREADS, row = c / BCPR, ch = c % BCPR;
    st_g[i] = row * BRB + ch * 16;
    st_r[i] = row * BRB + ((ch ^ u_swz(row)) << 4);
  }
  const int kv_end = CAUSAL ? min(p.Nk, q0 + BM + coff) : p.Nk;
  const int nT = (kv_end + BT - 1) / BT;

  constexpr bool DMA = FA_BWD_DMA != 0;
  static_assert(DMA || !PAD, "padded head dims are staged by LDS-DMA only");
  const unsigned dvo_ = PAD ? dma_off_pad(wave, lane) : dma_off(wave, lane);
  // (head_dim 256: a piece is 2 rows and a wave's pieces are 8 rows apart, half the swizzle's period: odd pieces flip bit 1 of the chunk)
  const unsigned dvo1 = dvo_ ^ 32u;
  // tile t -> buffer buf by LDS-DMA (hipcc does not count these loads: stage_write waits vmcnt(0))
  auto stage_dma = [&](int t, int buf) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const unsigned dvo = (BD == 256 && (j & 1)) ? dvo1 : dvo_;
      const unsigned soff = PAD ? (unsigned)(t * BT + j * 4 * RPP) * GRB : (unsigned)t * STILE + j * 4096;
      const unsigned lk = (unsigned)(__UINTPTR_TYPE__)KU + buf * STILE + (wave + 4 * j) * 1024;
      const unsigned lv = (unsigned)(__UINTPTR_TYPE__)VR + buf * STILE + (wave + 4 * j) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lk), "v"(dvo), "s"(rk), "s"(soff) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lv), "v"(dvo), "s"(rv), "s"(soff) : "memory");
    }
  };
  u32x4 kst[NCH], vst[NCH];
  auto stage_load = [&](int t, int buf) {
    if constexpr (DMA) {
      stage_dma(t, buf);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)t * STILE + st_g[i], 0, 0);
        vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)t * STILE + st_g[i], 0, 0);
      }
    }
  };
  auto stage_write = [&](int buf) {
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        lds_write_b128(KU + buf * STILE + st_r[i], kst[i]);
        lds_write_b128(VR + buf * STILE + st_r[i], vst[i]);
      }
    }
  };

  f32x16 dqacc[BDB];
#pragma unroll
  for (int db = 0; db < BDB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) dqacc[db][i] = 0.0f;

  stage_load(0, 0);
  stage_write(0);
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks)  // Q~ = round(c.Q): the very operand the forward multiplied (fa_mfma_kernel.hip)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[ks][j] = (elem)((float)qf[ks][j] * c2);
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(dof[ks]));  // retire the prologue loads
  __syncthreads();

  for (int t = 0; t < nT; ++t) {
    const int buf = t & 1;
    if (t + 1 < nT) stage_load(t + 1, buf ^ 1);
#pragma unroll
    for (int sub = 0; sub < BSUB; ++sub) {
    const int kv0 = t * BT + sub * BN;
    if (kv0 < kv_end && (!CAUSAL || kv0 <= qw0 + WM - 1 + coff)) {
      const int KS = sub * BTILE, VS = 2 * STILE + sub * BTILE;  // K / V sub-tile images, relative to koff / voff
      // masked: key > query (causal), and -- the partial last tile -- key >= Nk: those K / V rows arrive as zeros through the
      // descriptor, S' = -lse.log2e there, and with a strongly negative lse P = exp2(S') overflows the cast of dS (inf x 0 = NaN in dQ)
      const bool need_mask = (CAUSAL && (kv0 + BN - 1 > qw0 + coff)) || (kv0 + BN > p.Nk);
      // One 32-key half (kb) at a time -- scores, dS, then its share of dQ -- so that only one score and one dP tuple are live
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 sk, dpk;
        __builtin_amdgcn_s_setprio(1);  // matrix phases above the other wave's arithmetic (as in the forward kernel)
        {
          // 2 BKS row fragments (K and V alternating), each read LA products ahead of the MFMA that consumes it
          constexpr int NF = 2 * BKS, LA = FA_BWD_LA;
          vec8 fr[NF];
          auto fread = [&](int f) {  // f = (ks, which): which 0 = K row fragment, 1 = V row fragment
            fr[f] = __builtin_bit_cast(vec8, lds_read_b128(at(koff[f / 2] + ((f & 1) ? VS : KS) + kb * 32 * BRB)));
          };
#pragma unroll
          for (int f = 0; f < LA; ++f) fread(f);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int ks = f / 2;
            if (f & 1) dpk = M::mfma(fr[f], dof[ks], ks == 0 ? ndlt : dpk);
            else sk = M::mfma(fr[f], qf[ks], ks == 0 ? nlse : sk);
            if (f + LA < NF) fread(f + LA);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // dQ^T += K^T.dS^T : step j = (st, db); the transposed K fragments are read LA2 steps ahead of their MFMA, the first
        // ones before the dS arithmetic (they do not depend on it)
        constexpr int NJ = 2 * BDB, LA2 = FA_BWD_LA2, TV = BD == 64 ? 4 : 2;
        s16x4 tlo[NJ], thi[NJ];
        auto tread = [&](int j) {
          const int R0 = 32 * kb + 16 * (j / BDB), db = j % BDB;
          tlo[j] = lds_read_tr16(at(voff[(R0 >> 3) % TV][db] + KS + R0 * BRB));
          thi[j] = lds_read_tr16(at(voff[((R0 >> 3) + 1) % TV][db] + KS + (R0 + 8) * BRB));
        };
#pragma unroll
        for (int j = 0; j < LA2; ++j) tread(j);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        if (need_mask) {  // key > query -> masked (kernels.metal:748); a wave-uniform branch
          int lim = p.Nk - 1 - kv0 - 32 * kb - 4 * h;
          if (CAUSAL) lim = min(lim, qrow + coff - kv0 - 32 * kb - 4 * h);
#pragma unroll
          for (int i = 0; i < 16; ++i) sk[i] = ((i & 3) + 8 * (i >> 2) > lim) ? -INFINITY : sk[i];
        }
        // dS^T = P^T o (dP^T - delta) (the softmax scale goes onto the finished dQ): keys in the registers, the query on the lane
#pragma unroll
        for (int i = 0; i < 16; ++i) sk[i] = __builtin_amdgcn_exp2f(sk[i]) * dpk[i];
        vec8 df[2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) df[st][j] = (elem)sk[8 * st + j];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const s16x8 k8 = __builtin_shufflevector(tlo[j], thi[j], 0, 1, 2, 3, 4, 5, 6, 7);
          dqacc[j % BDB] = M::mfma(__builtin_bit_cast(vec8, k8), df[j / BDB], dqacc[j % BDB]);
          if (j + LA2 < NJ) tread(j + LA2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    }  // sub-tiles
#pragma unroll
    for (int ks = 0; ks < BKS; ++ks) {
      koff[ks] += flip;
      asm volatile("" : "+v"(koff[ks]));
    }
#pragma unroll
    for (int tv = 0; tv < NTV; ++tv)
#pragma unroll
      for (int db = 0; db < BDB; ++db) {
        voff[tv][db] += flip;
        asm volatile("" : "+v"(voff[tv][db]));
      }
    flip = -flip;
    if (t + 1 < nT) stage_write(buf ^ 1);
    __syncthreads();
  }
  // dQ^T[d][q]: lane (q = r, h) holds d = 32db + 8g4 + 4h + 0..3 -> one 16-byte store per group
  if (qvalid) {
    float *dq = p.dq + base + (long long)qrow * (PAD ? p.D : BD);
#pragma unroll
    for (int db = 0; db < BDB; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 w = make_float4(dqacc[db][4 * g4] * p.scale, dqacc[db][4 * g4 + 1] * p.scale, dqacc[db][4 * g4 + 2] * p.scale,
                                     dqacc[db][4 * g4 + 3] * p.scale);
        const int d0 = 32 * db + 8 * g4 + 4 * h;
        if (!PAD || d0 < p.D) *reinterpret_cast<float4 *>(dq + d0) = w;
      }
  }
}

// ---------------------------------------------------------------------------
// dK, dV: workgroup = 128 keys, wave = 32 keys (key on the lane, queries in the registers)
// ---------------------------------------------------------------------------
template <typename Tag, int D, bool CAUSAL, bool PAD>
__global__ __launch_bounds__(NTHREADS, bwd_occ(D)) void bwd_dkdv_kernel(BwdParams p) {
  FA_BWD_CONSTS(D, bwd_sub_kv(D));
  FA_BWD_PAD(PAD);
  using M = MT<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  lds_char *QU = smem;                // [2] Q tile (BT rows): read by rows for S, transposed for dK
  lds_char *OU = smem + 2 * STILE;    // [2] dO tile: read by rows for dP, transposed for dV
  lds_char *ROWS = smem + 4 * STILE;  // [2][2][BT] floats: -lse*log2e, -delta of the tile's query rows (the chains' initial accumulators)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // one workgroup per (key block, key/value head): it visits the G = H / Hkv query heads that read this head one after the
  // other, so grouped-query dK / dV are summed in registers (kernels.metal has one head count; G = 1 is its case)
  const int BHK = p.B * p.Hkv, G = p.H / p.Hkv;
  const int kvb = blockIdx.x / BHK;  // ascending: under the causal mask the first key blocks see the most queries
  const int bhk = blockIdx.x % BHK, bi = bhk / p.Hkv, hk = bhk % p.Hkv;
  const long long base = (long long)bi * p.kv_batch_stride + (long long)hk * p.kv_head_stride;  // K, V, dK, dV
  const int k0 = kvb * BM, kw0 = k0 + wave * WM, krow = kw0 + r;

  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const int coff = p.Nk - p.N;
  __amdgpu_buffer_rsrc_t rq, rdo;  // of the query head being visited (set_head)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.k + base), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.v + base), 0, kv_head_bytes, 0x00020000);

  vec8 kf[BKS], vf[BKS];  // B operands: lane (r,h) holds key row krow, columns 16ks+8h..
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) {
    kf[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)krow * GRB + gcol(2 * ks + h), 0, 0));
    vf[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)krow * GRB + gcol(2 * ks + h), 0, 0));
  }
  const float c2 = p.scale * LOG2E;

  const int kx = u_swz(r);
  const unsigned qu0 = (unsigned)(__UINTPTR_TYPE__)QU;
  unsigned koff[BKS];  // ABSOLUTE LDS addresses in the current Q buffer (the dO image is 2 STILE further)
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) koff[ks] = qu0 + r * BRB + (((2 * ks + h) ^ kx) << 4);
  const int g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  unsigned voff[NTV][BDB];
#pragma unroll
  for (int tv = 0; tv < NTV; ++tv)
#pragma unroll
    for (int db = 0; db < BDB; ++db) voff[tv][db] = qu0 + tr_off(tv, db, h, g1, vq, vp);
  // Offsets INCLUDING the current buffer's: with the buffer base added at the point of use hipcc kept a second, per-tile copy
  // of all twelve (base + offset) in registers (seen in the ISA, and spilled under the 168-register cap); they flip in place
  unsigned rowoff = (unsigned)(__UINTPTR_TYPE__)ROWS + 4 * h * 4;
  int flip = STILE, flip_rows = 2 * BT * 4;  // to the other buffer and back
  auto at = [](unsigned a) { return (const lds_char *)(__UINTPTR_TYPE__)a; };
  constexpr int NCH = BT * BCPR / NTHREADS;
  int st_g[NCH], st_r[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHREADS, row = c / BCPR, ch = c % BCPR;
    st_g[i] = row * BRB + ch * 16;
    st_r[i] = row * BRB + ((ch ^ u_swz(row)) << 4);
  }
  // query tiles of BT rows; under the causal mask only tiles that reach this block's first key
  const int nTq = (p.N + BT - 1) / BT;
  const int t_begin = CAUSAL ? max(k0 - coff, 0) / BT : 0;  // the first query that sees key k0 is k0 - coff
  static_assert(2 * BT <= NTHREADS, "one thread per staged row constant");

  u32x4 qst[NCH], ost[NCH];
  // threads 0..BT-1: lse of row tid of the next tile; BT..2BT-1: delta of row tid-BT. The RAW loaded value: any arithmetic on it
  // here makes hipcc wait for it -- vmcnt(0), i.e. for the whole tile's loads issued just before -- at the top of every
  // iteration (seen in the ISA: the memory latency was exposed once per tile). It is scaled / negated in stage_write.
  float rowv = 0.0f;
  // (2 BT threads = whole waves: the choice of array is wave-uniform and stays in scalar registers)
  const float *row_src = nullptr;
  auto set_head = [&](int g) {
    const int hq = hk * G + g;
    const long long bq = (long long)bi * p.batch_stride + (long long)hq * p.head_stride;
    rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.q + bq), 0, head_bytes, 0x00020000);
    rdo = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.d_o + bq), 0, head_bytes, 0x00020000);
    row_src = (__builtin_amdgcn_readfirstlane(tid) < BT ? p.lse : p.delta) + (long long)(bi * p.H + hq) * p.N;
  };
  constexpr bool DMA = FA_BWD_DMA != 0;
  static_assert(DMA || !PAD, "padded head dims are staged by LDS-DMA only");
  const unsigned dvo_ = PAD ? dma_off_pad(wave, lane) : dma_off(wave, lane);
  // (head_dim 256: a piece is 2 rows and a wave's pieces are 8 rows apart, half the swizzle's period: odd pieces flip bit 1 of the chunk)
  const unsigned dvo1 = dvo_ ^ 32u;
  auto stage_load = [&](int t, int buf) {
    if constexpr (DMA) {  // (hipcc does not count these loads: stage_write waits vmcnt(0))
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const unsigned dvo = (BD == 256 && (j & 1)) ? dvo1 : dvo_;
        const unsigned soff = PAD ? (unsigned)(t * BT + j * 4 * RPP) * GRB : (unsigned)t * STILE + j * 4096;
        const unsigned lq = (unsigned)(__UINTPTR_TYPE__)QU + buf * STILE + (wave + 4 * j) * 1024;
        const unsigned lo = (unsigned)(__UINTPTR_TYPE__)OU + buf * STILE + (wave + 4 * j) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lq), "v"(dvo), "s"(rq), "s"(soff) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lo), "v"(dvo), "s"(rdo), "s"(soff) : "memory");
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        qst[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)t * STILE + st_g[i], 0, 0);
        ost[i] = __builtin_amdgcn_raw_buffer_load_b128(rdo, (unsigned)t * STILE + st_g[i], 0, 0);
      }
    }
    if (tid < 2 * BT) {
      const int qi = t * BT + (tid & (BT - 1));
      rowv = row_src[qi < p.N ? qi : p.N - 1];
    }
  };
  auto stage_write = [&](int buf, int wt) {  // wt = the tile the staged registers hold
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        lds_write_b128(QU + buf * STILE + st_r[i], qst[i]);
        lds_write_b128(OU + buf * STILE + st_r[i], ost[i]);
      }
    }
    if (tid < 2 * BT) {  // (the staged registers hold tile wt: its rows past N get p = 0 through -inf)
      const int qi = wt * BT + (tid & (BT - 1));
      const float v = (tid < BT) ? (qi < p.N ? -rowv * LOG2E : -INFINITY) : (qi < p.N ? -rowv : 0.0f);
      lds_write_b32(ROWS + buf * (2 * BT * 4) + tid * 4, __builtin_bit_cast(unsigned, v));
    }
  };

  f32x16 dkacc[BDB], dvacc[BDB];
#pragma unroll
  for (int db = 0; db < BDB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dkacc[db][i] = 0.0f; dvacc[db][i] = 0.0f; }

  set_head(0);
  if (t_begin < nTq) {
    stage_load(t_begin, 0);
    stage_write(0, t_begin);
  }
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks)  // K~ = round(c.K): S' = Q.K~ comes out in log2 units
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[ks][j] = (elem)((float)kf[ks][j] * c2);
#pragma unroll
  for (int ks = 0; ks < BKS; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
  __syncthreads();

  int buf = 0;  // the buffer koff / voff / rowoff point into
  for (int g = 0; g < G; ++g) {
  if (g > 0) {  // next query head of the group (every wave is past the last tile's barrier: both buffers are free)
    set_head(g);
    if (t_begin < nTq) {
      stage_load(t_begin, buf);
      stage_write(buf, t_begin);
    }
    __syncthreads();
  }
  for (int t = t_begin; t < nTq; ++t) {
    if (t + 1 < nTq) stage_load(t + 1, buf ^ 1);
#pragma unroll
    for (int sub = 0; sub < BSUB; ++sub) {
    const int sub_c = sub;
    const int qt0 = t * BT + sub * BN;
    if (qt0 < p.N && (!CAUSAL || qt0 + BN - 1 + coff >= kw0)) {  // some query of the sub-tile sees this wave's first key
      // (the buffer's offset is inside koff / voff / rowoff, toggled once per tile: everything added here is an immediate)
      const int QS = sub_c * BTILE, OS = 2 * STILE + sub_c * BTILE;  // Q / dO sub-tile images, relative to koff / voff
      const unsigned rows = rowoff + sub_c * (BN * 4);
      // only sub-tiles that cross the diagonal for this wave need the per-element mask (wave-uniform)
      const bool need_mask = CAUSAL && (qt0 + coff < kw0 + WM - 1);
      // One 32-query half (qb) at a time -- scores, P / dS, then its share of dV / dK -- so that only ONE score and ONE dP tuple
      // are live (round 3 first kept both halves': 212 VGPR, two waves per SIMD; this form fits three).
      static_for<0, 2>([&](auto qbc) {
        constexpr int qb = decltype(qbc)::value;
        f32x16 sq, dpq;
        __builtin_amdgcn_s_setprio(1);  // matrix phases above the other wave's arithmetic (as in the forward kernel)
        // the chains start from the row constants: registers 4g..4g+3 are query rows 32qb + 8g + 4h + 0..3 of the sub-tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ql = 32 * qb + 8 * g;  // (+ 4h: in rowoff)
          const u32x4 l4 = lds_read_b128(at(rows + ql * 4));
          const u32x4 d4 = lds_read_b128(at(rows + BT * 4 + ql * 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (scalar temporaries on purpose: __builtin_bit_cast applied directly to the vector element expression
            //  l4[e] read element 0 for every e -- seen in the IR)
            const unsigned lw = l4[e], dw = d4[e];
            sq[4 * g + e] = __builtin_bit_cast(float, lw);
            dpq[4 * g + e] = __builtin_bit_cast(float, dw);
          }
        }
        {
          constexpr int NF = 2 * BKS, LA = BD >= 128 ? FA_BWD_KV128_LA : FA_BWD_LA;
          vec8 fr[NF];
          auto fread = [&](auto fc) {  // f = (ks, which): which 0 = Q row fragment, 1 = dO row fragment
            constexpr int f = decltype(fc)::value;
            fr[f] = __builtin_bit_cast(
                vec8, lds_read_b128(at(koff[f / 2] + ((f & 1) ? OS : QS) + qb * 32 * BRB)));
          };
          static_for<0, LA>([&](auto fc) { fread(fc); });
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f & 1) dpq = M::mfma(fr[f], vf[f / 2], dpq);
            else sq = M::mfma(fr[f], kf[f / 2], sq);
            if constexpr (f + LA < NF) fread(std::integral_constant<int, f + LA>{});
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        // dV / dK fragments of this half (step j = (st, db, which): which 0 = dO^T fragment -> dV, 1 = Q^T fragment -> dK) are
        // read LA2 steps ahead of their MFMA, the first ones before the P / dS arithmetic (they do not depend on it)
        constexpr int NJ = 4 * BDB, LA2 = BD >= 128 ? FA_BWD_KV128_LA2 : FA_BWD_LA2, TV = BD == 64 ? 4 : 2;
        s16x4 tlo[NJ], thi[NJ];
        auto tread = [&](auto jc) {
          constexpr int j = decltype(jc)::value, jj = j / 2, R0 = 32 * qb + 16 * (jj / BDB), db = jj % BDB;
          const int src = (j & 1) ? QS : OS;
          tlo[j] = lds_read_tr16(at(voff[(R0 >> 3) % TV][db] + src + R0 * BRB));
          thi[j] = lds_read_tr16(at(voff[((R0 >> 3) + 1) % TV][db] + src + (R0 + 8) * BRB));
        };
        static_for<0, LA2>([&](auto jc) { tread(jc); });
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        if (need_mask) {  // key > query (kernels.metal:748): S' = -inf there. A wave-uniform BRANCH: written as a per-element
          // condition hipcc turned it into 32 compare + select pairs on every tile (seen in the ISA). Register 4g+e holds query
          // qt0 + 32qb + 8g + 4h + e: compared as a constant against ONE per-lane limit (else: sixteen threshold registers)
          const int lim = krow - coff - 4 * h - qt0 - 32 * qb;
#pragma unroll
          for (int i = 0; i < 16; ++i) sq[i] = (8 * (i >> 2) + (i & 3) < lim) ? -INFINITY : sq[i];
        }
        vec8 pf[2], df[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pv = __builtin_amdgcn_exp2f(sq[i]);  // S' = Q.K~ - lse*log2e came out of the matrix core
          sq[i] = pv;
          dpq[i] = pv * dpq[i];  // dS (without the softmax scale: it goes onto the finished dK)
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pf[st][j] = (elem)sq[8 * st + j];
            df[st][j] = (elem)dpq[8 * st + j];
          }
        // dV^T += dO^T.P ; dK^T += Q^T.dS   (reduction over this half's 32 query rows)
        __builtin_amdgcn_s_setprio(1);
        static_for<0, NJ>([&](auto jc) {
          constexpr int j = decltype(jc)::value, jj = j / 2, st = jj / BDB, db = jj % BDB;
          const s16x8 a8 = __builtin_shufflevector(tlo[j], thi[j], 0, 1, 2, 3, 4, 5, 6, 7);
          if constexpr (j & 1) dkacc[db] = M::mfma(__builtin_bit_cast(vec8, a8), df[st], dkacc[db]);
          else dvacc[db] = M::mfma(__builtin_bit_cast(vec8, a8), pf[st], dvacc[db]);
          if constexpr (j + LA2 < NJ) tread(std::integral_constant<int, j + LA2>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
    }  // sub-tiles
#pragma unroll
    for (int ks = 0; ks < BKS; ++ks) {
      koff[ks] += flip;
      asm volatile("" : "+v"(koff[ks]));
    }
#pragma unroll
    for (int tv = 0; tv < NTV; ++tv)
#pragma unroll
      for (int db = 0; db < BDB; ++db) {
        voff[tv][db] += flip;
        asm volatile("" : "+v"(voff[tv][db]));
      }
    rowoff += flip_rows;
    asm volatile("" : "+v"(rowoff));
    flip = -flip;
    flip_rows = -flip_rows;
    if (t + 1 < nTq) stage_write(buf ^ 1, t + 1);
    __syncthreads();
    buf ^= 1;
  }
  }  // query heads of the group
  if (krow < p.Nk) {
    float *dk = p.dk + base + (long long)krow * (PAD ? p.D : BD), *dv = p.dv + base + (long long)krow * (PAD ? p.D : BD);
#pragma unroll
    for (int db = 0; db < BDB; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 32 * db + 8 * g4 + 4 * h;
        if (PAD && d0 >= p.D) continue;
        *reinterpret_cast<float4 *>(dk + d0) = make_float4(dkacc[db][4 * g4] * p.scale, dkacc[db][4 * g4 + 1] * p.scale, dkacc[db][4 * g4 + 2] * p.scale,
                                                             dkacc[db][4 * g4 + 3] * p.scale);
        *reinterpret_cast<float4 *>(dv + d0) = make_float4(dvacc[db][4 * g4], dvacc[db][4 * g4 + 1], dvacc[db][4 * g4 + 2], dvacc[db][4 * g4 + 3]);
      }
  }
}

// ---------------------------------------------------------------------------
bool bwd_supported(int dtype, int D) {
  if (dtype == FA_DTYPE_FP8_E4M3) return D >= 16 && D <= 128 && D % 16 == 0;  // (rows of whole 16-byte chunks: the widening pass)
  return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16) && ((D >= 8 && D <= 128 && D % 8 == 0) || D == 256);
}

// e4m3 inputs (the forward's config-5 family): Q, K, V are widened to bf16 -- exactly: every e4m3 value is a bf16 value -- into the
// caller's workspace by one streaming pass, under their own element strides, and the bf16 kernels run on the copies (O and dO of that
// family are bf16 already). 3 bytes of traffic per element against the backward's hundreds of FLOPs per element.
__global__ __launch_bounds__(256) void widen_e4m3_kernel(const unsigned *in, unsigned *out, long long n16) {
  // one thread = 16 elements: 16 bytes in, 32 bytes out
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    const u32x4 w = *reinterpret_cast<const u32x4 *>(in + 4 * i);
    u32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[j], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[j], true);
      // bf16 = the upper half of the fp32 pattern (exact here: an e4m3 value has 3 mantissa bits)
      // (scalar temporaries on purpose: __builtin_bit_cast applied directly to a vector element expression reads element 0, see below)
      const float a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
      const unsigned p0 = (__builtin_bit_cast(unsigned, a0) >> 16) | (__builtin_bit_cast(unsigned, a1) & 0xffff0000u);
      const unsigned p1 = (__builtin_bit_cast(unsigned, b0) >> 16) | (__builtin_bit_cast(unsigned, b1) & 0xffff0000u);
      if (j < 2) { lo[2 * j] = p0; lo[2 * j + 1] = p1; } else { hi[2 * (j - 2)] = p0; hi[2 * (j - 2) + 1] = p1; }
    }
    *reinterpret_cast<u32x4 *>(out + 8 * i) = lo;
    *reinterpret_cast<u32x4 *>(out + 8 * i + 4) = hi;
  }
}

// n elements (a multiple of 16, 16-byte aligned source) of e4m3 -> bf16
hipError_t launch_widen_e4m3(const void *in, void *out, long long n, hipStream_t s) {
  const long long n16 = n / 16;
  long long blocks = (n16 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(widen_e4m3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned *)in, (unsigned *)out, n16);
  return hipGetLastError();
}

template <typename Tag, int D, bool CAUSAL, bool PAD>
static hipError_t launch_bwd_one(const BwdParams &p, hipStream_t s) {
  constexpr int BTILE = BN * D * 2;
  const int nBq = (p.N + BM - 1) / BM, nBk = (p.Nk + BM - 1) / BM;
  const size_t smem_dq = 4 * bwd_sub_dq(D) * BTILE, smem_kv = 4 * bwd_sub_kv(D) * BTILE + 2 * 2 * bwd_sub_kv(D) * BN * 4;
  auto kq = bwd_dq_kernel<Tag, D, CAUSAL, PAD>;
  auto kk = bwd_dkdv_kernel<Tag, D, CAUSAL, PAD>;
  hipError_t e = hipSuccess;
  if (smem_kv > 48 * 1024) e = set_dyn_lds_once((const void *)kk, (int)smem_kv);
  if (e != hipSuccess) return e;
  if (smem_dq > 48 * 1024) {
    e = set_dyn_lds_once((const void *)kq, (int)smem_dq);
    if (e != hipSuccess) return e;
  }
  (void)hipGetLastError();  // do not report an older sticky error as this launch's
  hipLaunchKernelGGL(kq, dim3(nBq * p.B * p.H), dim3(NTHREADS), smem_dq, s, p);
  hipLaunchKernelGGL(kk, dim3(nBk * p.B * p.Hkv), dim3(NTHREADS), smem_kv, s, p);
  return hipGetLastError();
}

template <typename Tag>
static hipError_t launch_bwd_dt(const BwdParams &p, hipStream_t s) {
  if (p.D == 64) return p.is_causal ? launch_bwd_one<Tag, 64, true, false>(p, s) : launch_bwd_one<Tag, 64, false, false>(p, s);
  if (p.D == 128) return p.is_causal ? launch_bwd_one<Tag, 128, true, false>(p, s) : launch_bwd_one<Tag, 128, false, false>(p, s);
  if (p.D == 256) return p.is_causal ? launch_bwd_one<Tag, 256, true, false>(p, s) : launch_bwd_one<Tag, 256, false, false>(p, s);
  if (p.D < 64) return p.is_causal ? launch_bwd_one<Tag, 64, true, true>(p, s) : launch_bwd_one<Tag, 64, false, true>(p, s);
  return p.is_causal ? launch_bwd_one<Tag, 128, true, true>(p, s) : launch_bwd_one<Tag, 128, false, true>(p, s);
}

hipError_t launch_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse,
                      float *dq, float *dk, float *dv, float *ws, int B, int H, int Hkv, int N, int Nk, int D, float scale,
                      long long bs, long long hs, long long kv_bs, long long kv_hs, int causal, int dtype, hipStream_t s) {
  BwdParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.lse = lse;
  p.dq = dq; p.dk = dk; p.dv = dv; p.delta = ws;
  p.B = B; p.H = H; p.N = N; p.D = D; p.scale = scale;
  p.batch_stride = bs; p.head_stride = hs; p.is_causal = causal;
  p.Hkv = Hkv; p.Nk = Nk; p.kv_batch_stride = kv_bs; p.kv_head_stride = kv_hs;
  return dtype == FA_DTYPE_F16 ? launch_bwd_dt<F16>(p, s) : launch_bwd_dt<BF16>(p, s);
}

}  // namespace fa
