// fa_fwd_splitkv_kernel.hip -- the operator for SMALL grids: the keys of one 32-row query block are split
// over the waves of a workgroup and merged in LDS.
//
// Why: the metric is TFLOPS and ms *versus sequence length* (/root/reference/main.mm:608 sweeps 128...16384).
// With few (batch, head) slices and a short sequence the 128-row kernel launches fewer workgroups than the
// chip has CUs and each of them walks its KV tiles one after another: config 2 (8 heads x 1024 keys) is 64
// workgroups x 16 tiles = 16-19 us against 0.9 us of arithmetic. Here a workgroup is ONE 32-row query
// block and its S = 2, 4 or 8 waves each take every S-th KV tile (strided, so causal blocks stay
// balanced), with their own (m, l, O^T) and their own LDS tile -- no barrier in the loop, each wave runs
// at its own pace -- and one merge at the end:
//     M = max_w m_w,   l = sum_w l_w 2^(c (m_w - M)),   O = sum_w O_w 2^(c (m_w - M)) / l,   LSE = M.scale + ln l
// (the LSE of /root/reference/kernels.metal:862-864 is exactly what makes partial results mergeable).
// No global workspace: the C-ABI promises that the library allocates nothing (include/fa_mi355.h).
//
// Same operand maps, LDS images, mask predicate and deferred-max rule as fa_mfma_kernel.hip.
#include "fa_mfma_common.h"

#ifndef FA_DEFER_THR
#define FA_DEFER_THR 8.0f
#endif

namespace fa {

template <typename Tag, int D, bool CAUSAL>
// (head_dim 128 never runs more than four waves -- splitkv_waves: LDS -- so its register cap is 512, not 256: compiled for
//  eight it spilled 820 B per lane and ran 3-4x slower than the kernels it was chosen over, profiles/r03/ab_d128_small_grids.log)
__global__ __launch_bounds__((D == 64 ? 512 : 256), 1) void fwd_splitkv_kernel(Params p) {
  using M = MT<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  constexpr int RB = D * 2;
  constexpr int KS = D / 16;
  constexpr int DB = D / 32;
  constexpr int TILE = BN * RB;               // one K (or V) tile in LDS
  constexpr bool IS_FP8 = std::is_same<Tag, FP8>::value;
  constexpr int GB = IS_FP8 ? 1 : 2;
  constexpr int GRB = D * GB;
  constexpr int GTILE = BN * GRB;
  constexpr int GCPR = GRB / 16;
  constexpr int NCH = BN * GCPR / 64;         // 16-byte global chunks per LANE per tile (the wave stages its own tiles)
  constexpr int NACC = 16 * DB;               // O^T registers per lane

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  const int S = blockDim.x >> 6;              // waves = KV splits
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  lds_char *Kt = smem + wave * (2 * TILE);    // this wave's private K tile, V tile behind it
  lds_char *Vt = Kt + TILE;

  const int nQ = p.nq;
  const int bh = (int)fdiv(blockIdx.x, p.fd_nq), qrem = (int)blockIdx.x - bh * nQ;
  const int qb = CAUSAL ? (nQ - 1 - qrem) : qrem;  // causal: heavy blocks first
  long long base, base_kv;
  head_bases(bh, p, base, base_kv);
  const int coff = p.Nk - p.N;
  const int q0 = qb * WM;
  const int qrow = q0 + r;

  const unsigned head_bytes = (unsigned)p.N * GRB, kv_head_bytes = (unsigned)p.Nk * GRB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.q + base * GB), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.k + base_kv * GB), 0, kv_head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)p.v + base_kv * GB), 0, kv_head_bytes, 0x00020000);

  vec8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if constexpr (IS_FP8) {
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rq, (unsigned)qrow * GRB + (2 * ks + h) * 8, 0, 0);
      qf[ks] = __builtin_bit_cast(vec8, fp8x8_to_bf16(t));
    } else {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * RB + (2 * ks + h) * 16, 0, 0);
      qf[ks] = __builtin_bit_cast(vec8, t);
    }
  }

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

  // staging map of ONE wave: lane -> NCH chunks; chunk c = lane + 64 i: row c / GCPR, global chunk c % GCPR
  const int row0 = lane / GCPR, gch = lane % GCPR;
  constexpr int ROWS_PER_I = 64 / GCPR;  // chunk i sits ROWS_PER_I rows below chunk i-1
  const int st_g0 = row0 * GRB + gch * 16;
  const int chl = IS_FP8 ? 2 * gch : gch;
  auto k_slot = [&](int row, int ch) __attribute__((always_inline)) {  // LDS byte offset of 16-byte chunk ch of K row `row`
    const int x = (D == 64) ? ((row >> 1) & 7) : (row & 15);
    return row * RB + ((ch ^ x) << 4);
  };
  auto v_slot = [&](int row, int ch) __attribute__((always_inline)) {
    const int x = (D == 64) ? (((row >> 1) & 1) << 2) : ((row & 3) << 2);
    return row * RB + ((ch ^ x) << 4);
  };

  const int kv_end = CAUSAL ? min(p.Nk, q0 + WM + coff) : p.Nk;
  const int nT = (kv_end + BN - 1) / BN;  // tiles of this query block; wave w takes w, w+S, ...

  // two staging register sets: a wave's first TWO tiles are requested together with Q (short sequences leave
  // a wave only a few tiles, so every exposed memory latency counts), later ones two tiles ahead
  u32x4 kstA[NCH], vstA[NCH], kstB[NCH], vstB[NCH];
  auto stage_load = [&](u32x4 (&kst)[NCH], u32x4 (&vst)[NCH], int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const unsigned g = (unsigned)t * GTILE + st_g0 + i * ROWS_PER_I * GRB;
      kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, g, 0, 0);
      vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, g, 0, 0);
    }
  };
  auto stage_write = [&](u32x4 (&kst)[NCH], u32x4 (&vst)[NCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + i * ROWS_PER_I;
      if constexpr (IS_FP8) {
        lds_write_b128(Kt + k_slot(row, chl), fp8x8_to_bf16(u32x2{kst[i][0], kst[i][1]}));
        lds_write_b128(Kt + k_slot(row, chl + 1), fp8x8_to_bf16(u32x2{kst[i][2], kst[i][3]}));
        lds_write_b128(Vt + v_slot(row, chl), fp8x8_to_bf16(u32x2{vst[i][0], vst[i][1]}));
        lds_write_b128(Vt + v_slot(row, chl + 1), fp8x8_to_bf16(u32x2{vst[i][2], vst[i][3]}));
      } else {
        lds_write_b128(Kt + k_slot(row, chl), kst[i]);
        lds_write_b128(Vt + v_slot(row, chl), vst[i]);
      }
    }
  };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.0f;
  float m = -INFINITY, mthr = -INFINITY, l = 0.0f;
  const float c2 = p.scale * 1.4426950408889634f;
  const float thr_raw = FA_DEFER_THR / c2;

  auto tile = [&](const int t) __attribute__((always_inline)) {
    const int kv0 = t * BN;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[kb] = M::mfma(__builtin_bit_cast(vec8, lds_read_b128(Kt + kb * 32 * RB + koff[ks])), qf[ks], s[kb]);
    }
    const bool need_mask = (CAUSAL && (kv0 + BN - 1 > q0 + coff)) || (kv0 + BN > p.Nk);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        int lim = p.Nk - 1 - kv0 - 32 * kb - 4 * h;  // masked iff key > query (kernels.metal:748) or key >= Nk
        if (CAUSAL) lim = min(lim, qrow + coff - kv0 - 32 * kb - 4 * h);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kpart = (i & 3) + 8 * (i >> 2);
          s[kb][i] = (kpart > lim) ? -INFINITY : s[kb][i];
        }
      }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);
    {
      float lo, hi;
      half_pair(mx, lo, hi);
      mx = fmaxf(lo, hi);
    }
    if (__builtin_amdgcn_ballot_w64(mx > mthr) != 0) {
      const float m_new = fmaxf(m, mx);
      // a wave's FIRST tile can be entirely masked for some rows (rectangular causal shapes): the reference stays
      // -inf there and the row contributes nothing (alpha below: exp2 of -inf - -inf must not be evaluated)
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m - m_new) * c2);
      l *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
      m = m_new;
      mthr = m_new + thr_raw;
    }
    const float mc = (m == -INFINITY) ? 0.0f : m * c2;  // all scores of such a row are -inf: exp2(-inf - 0) = 0
    float ls0 = 0.0f, ls1 = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[0][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][i], c2, -mc));
      s[1][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[1][i], c2, -mc));
      ls0 += s[0][i];
      ls1 += s[1][i];
    }
    l += ls0 + ls1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        vec8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (elem)s[kb][8 * st + j];
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const lds_char *vb = Vt + (32 * kb + 16 * st) * RB + voff[db];
          const s16x8 v8 = __builtin_shufflevector(lds_read_tr16(vb), lds_read_tr16(vb + 8 * RB), 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[db] = M::mfma(__builtin_bit_cast(vec8, v8), pf, oacc[db]);
        }
      }
  };
  if (wave < nT) stage_load(kstA, vstA, wave);
  if (wave + S < nT) stage_load(kstB, vstB, wave + S);
  for (int t = wave; t < nT; t += 2 * S) {
    stage_write(kstA, vstA);  // this wave's LDS tile: its own stores, its own reads -- program order suffices
    if (t + 2 * S < nT) stage_load(kstA, vstA, t + 2 * S);
    tile(t);
    if (t + S < nT) {
      stage_write(kstB, vstB);
      if (t + 3 * S < nT) stage_load(kstB, vstB, t + 3 * S);
      tile(t + S);
    }
  }

  // ---- merge, all waves at once: every wave publishes (m, l, O^T) lane by lane; after ONE barrier wave w folds the
  // registers e = w, w+S, ... of all S partial results (each lane reads exactly the elements of its own (row, d)
  // positions: conflict-free), normalises them and stores them -- 4 consecutive head-dim elements per lane and register
  // group. (A serial merge by wave 0 cost 7 x 34 LDS reads x 64 lanes in a row: 2 us of a 10 us kernel.)
  {
    float lo, hi;
    half_pair(l, lo, hi);
    l = lo + hi;  // whole-row sum of this wave's share (both lane halves hold it now)
  }
  __syncthreads();  // every wave is done with its K/V tile: the LDS becomes the merge buffer
  float *mb = (float *)smem_generic;  // [S][NACC + 2][64] floats: element e of wave w, lane at [w (NACC+2) + e][lane]
  {
    float *dst = mb + (size_t)wave * (NACC + 2) * 64 + lane;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[(16 * db + i) * 64] = oacc[db][i];
    dst[NACC * 64] = m;
    dst[(NACC + 1) * 64] = l;
  }
  __syncthreads();
  float Mx = -INFINITY;
  for (int w = 0; w < S; ++w) Mx = fmaxf(Mx, mb[((size_t)w * (NACC + 2) + NACC) * 64 + lane]);  // wave 0 owns tile 0: Mx is finite
  float lt = 0.0f;
  float aw[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    aw[w] = 0.0f;
    if (w < S) {
      const float *src = mb + (size_t)w * (NACC + 2) * 64 + lane;
      aw[w] = __builtin_amdgcn_exp2f((src[NACC * 64] - Mx) * c2);  // a share without a visible key has m = -inf -> weight 0
      lt += src[(NACC + 1) * 64] * aw[w];
    }
  }
  const float inv_l = 1.0f / lt;
  if (wave == 0 && p.lse != nullptr && h == 0 && qrow < p.N) p.lse[(long long)bh * p.N + qrow] = Mx * p.scale + logf(lt);
  // register groups of 4 (= 4 consecutive head-dim elements of one row): group g = 4 registers 4g .. 4g+3; wave w takes g = w, w+S, ...
  elem *Og = (elem *)p.o + base;
  for (int g = wave; g < NACC / 4; g += S) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < 8; ++w)
      if (w < S) {
        const float *src = mb + ((size_t)w * (NACC + 2) + 4 * g) * 64 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += src[j * 64] * aw[w];
      }
    typedef elem elem2 __attribute__((ext_vector_type(2)));
    elem2 a, b;
    a[0] = (elem)(acc[0] * inv_l);
    a[1] = (elem)(acc[1] * inv_l);
    b[0] = (elem)(acc[2] * inv_l);
    b[1] = (elem)(acc[3] * inv_l);
    u32x2 wv;
    wv[0] = __builtin_bit_cast(unsigned, a);
    wv[1] = __builtin_bit_cast(unsigned, b);
    const int db = g / 4, g4 = g % 4;
    const int col = 32 * db + 8 * g4 + 4 * h;  // registers 4g4..4g4+3 of tuple db = head-dim columns col..col+3 of row r
    if (qrow < p.N) *reinterpret_cast<u32x2 *>(Og + (long long)qrow * D + col) = wv;
  }
}

// ---------------------------------------------------------------------------
bool splitkv_supported(int dtype, int D) {
  // head_dim 64 only since round 4: the head_dim-128 instantiation spilled 172 B under its four-wave register cap, lost to the
  // eight-wave form on every small grid (profiles/r03/ab_d128_small_grids.log) and was reachable by name only
  return (dtype == FA_DTYPE_F16 || dtype == FA_DTYPE_BF16 || dtype == FA_DTYPE_FP8_E4M3) && D == 64;
}

// waves per workgroup = KV splits: the largest of 8, 4, 2 that leaves every wave a tile and fits the LDS
int splitkv_waves(int D, int Nk) {
  const int nT = (Nk + BN - 1) / BN;
  const int max_by_lds = (D == 64) ? 8 : 4;  // 2 tiles of 64 x D x 2 bytes per wave, 160 KiB per CU
  int S = 8;
  while (S > 2 && (S > nT || S > max_by_lds)) S >>= 1;
  return S;
}

template <typename Tag, int D, bool CAUSAL>
static hipError_t launch_splitkv_one(const Params &p, hipStream_t s) {
  const int S = splitkv_waves(D, p.Nk);
  const int nQ = (p.N + WM - 1) / WM;
  size_t smem = (size_t)S * 2 * BN * D * 2;                                // S private K+V tiles ...
  const size_t merge = (size_t)S * (16 * (D / 32) + 2) * 64 * 4;           // ... reused as the merge buffer
  if (merge > smem) smem = merge;
  auto kern = fwd_splitkv_kernel<Tag, D, CAUSAL>;
  if (smem > 48 * 1024) {
    hipError_t e = set_dyn_lds_once((const void *)kern, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  // one workgroup per 32-row block: four times the grid fa_fwd's "grid too large" guard (128-row blocks) lets through
  const long long grid = (long long)nQ * p.B * p.H;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  Params pp = p;
  set_block_divisors(pp, nQ, 0);
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * S), smem, s, pp);
  return hipGetLastError();
}

template <typename Tag>
static hipError_t launch_splitkv_dt(const Params &p, hipStream_t s) {
  if (p.D == 64) return p.is_causal ? launch_splitkv_one<Tag, 64, true>(p, s) : launch_splitkv_one<Tag, 64, false>(p, s);
  return hipErrorInvalidValue;
}

hipError_t launch_splitkv(const Params &p, int dtype, hipStream_t s) {
  if (dtype == FA_DTYPE_FP8_E4M3) return launch_splitkv_dt<FP8>(p, s);
  return dtype == FA_DTYPE_F16 ? launch_splitkv_dt<F16>(p, s) : launch_splitkv_dt<BF16>(p, s);
}

}  // namespace fa
