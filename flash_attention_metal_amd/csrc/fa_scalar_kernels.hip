// fa_scalar_kernels.hip -- the three non-matrix-core variants of the operator
// for gfx950 (wave64). They fill the reference's CSV columns 2-4 and serve as
// the fp32 device baseline the reference's own checks are anchored on:
//
//   naive     one thread per query row, two passes over K
//             (semantics of /root/reference/kernels.metal:12-64)
//   tiled     K/V tiles staged in LDS, per-key online softmax, scalar loads
//             (semantics of kernels.metal:72-171, "V1")
//   tiled_v2  128-bit global loads, K/V double-buffered in LDS with register
//             prefetch, one barrier per tile, per-tile online softmax
//             (semantics of kernels.metal:462-596, "V2"; also 16-bit I/O for
//             BASELINE.json config 2)
//
// All three carry the full operator signature (batch/head strides, is_causal,
// LSE) which the reference only gives its V4 kernel. fp32 accumulation always.
// Written for CDNA4: a wave is 64 lanes, so a row-per-lane block is a multiple
// of 64 rows; K/V rows are read from LDS at one address per wave-instruction
// (hardware broadcast, conflict-free).
#include <mutex>

#include "fa_common.h"

namespace fa {

template <typename Tag> struct Elem;
template <> struct Elem<F32> { using type = float; };
template <> struct Elem<F16> { using type = _Float16; };
template <> struct Elem<BF16> { using type = __bf16; };

// ---------------------------------------------------------------------------
// naive: kernels.metal:12-64
// ---------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void naive_kernel(Params p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.N) return;
  const long long base = (long long)blockIdx.z * p.batch_stride + (long long)blockIdx.y * p.head_stride;
  const T *Q = (const T *)p.q + base, *K = (const T *)p.k + base, *V = (const T *)p.v + base;
  T *O = (T *)p.o + base;
  const int jend = p.is_causal ? row + 1 : p.N;

  float qreg[D];
#pragma unroll
  for (int d = 0; d < D; ++d) qreg[d] = ld_elem(Q, (long long)row * D + d);

  // pass 1: max (kernels.metal:35-43)
  float max_score = -INFINITY;
  for (int j = 0; j < jend; ++j) {
    float score = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) score += qreg[d] * ld_elem(K, (long long)j * D + d);
    score *= p.scale;
    if (score > max_score) max_score = score;
  }
  // pass 2: exp, sum, weighted V (kernels.metal:46-58)
  float acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.0f;
  float sum_exp = 0.0f;
  for (int j = 0; j < jend; ++j) {
    float score = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) score += qreg[d] * ld_elem(K, (long long)j * D + d);
    score *= p.scale;
    const float pj = expf(score - max_score);
    sum_exp += pj;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += pj * ld_elem(V, (long long)j * D + d);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) st_elem(O, (long long)row * D + d, acc[d] / sum_exp);
  if (p.lse) {
    const long long bh = (long long)blockIdx.z * p.H + blockIdx.y;
    p.lse[bh * p.N + row] = max_score + logf(sum_exp);
  }
}

// ---------------------------------------------------------------------------
// tiled ("V1"): kernels.metal:72-171. One wave = 64 query rows, Bc = 32.
// ---------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(64) void tiled_kernel(Params p) {
  constexpr int BR = 64, BC = 32;
  __shared__ float Ks[BC * D];
  __shared__ float Vs[BC * D];
  const int tx = threadIdx.x;
  const int row0 = blockIdx.x * BR;
  const int row = row0 + tx;
  const bool valid = row < p.N;
  const long long base = (long long)blockIdx.z * p.batch_stride + (long long)blockIdx.y * p.head_stride;
  const T *Q = (const T *)p.q + base, *K = (const T *)p.k + base, *V = (const T *)p.v + base;
  T *O = (T *)p.o + base;

  float qreg[D], acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qreg[d] = valid ? ld_elem(Q, (long long)row * D + d) : 0.0f;
    acc[d] = 0.0f;
  }
  float m = -INFINITY, l = 0.0f;

  // whole-tile causal skip: the analogue of kernels.metal:682
  const int last_row = min(row0 + BR, p.N) - 1;
  const int kv_end = p.is_causal ? last_row + 1 : p.N;
  for (int kv0 = 0; kv0 < kv_end; kv0 += BC) {
    for (int idx = tx; idx < BC * D; idx += BR) {  // scalar, zero-padded (kernels.metal:127-132)
      const int r = idx / D, c = idx % D;
      const bool in = kv0 + r < p.N;
      Ks[idx] = in ? ld_elem(K, (long long)(kv0 + r) * D + c) : 0.0f;
      Vs[idx] = in ? ld_elem(V, (long long)(kv0 + r) * D + c) : 0.0f;
    }
    __syncthreads();
    const int jn = min(BC, p.N - kv0);
    for (int j = 0; j < jn; ++j) {
      const int key = kv0 + j;
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += qreg[d] * Ks[j * D + d];
      s *= p.scale;
      if (!p.is_causal || key <= row) {  // kernels.metal:748 predicate, per element
        // per-key online update (kernels.metal:140-160)
        const float m_new = fmaxf(m, s);
        const float alpha = expf(m - m_new);
        const float pj = expf(s - m_new);
        l = l * alpha + pj;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * alpha + pj * Vs[j * D + d];
        m = m_new;
      }
    }
    __syncthreads();
  }
  if (!valid) return;
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < D; ++d) st_elem(O, (long long)row * D + d, acc[d] * inv);
  if (p.lse) {
    const long long bh = (long long)blockIdx.z * p.H + blockIdx.y;
    p.lse[bh * p.N + row] = m + logf(l);
  }
}

// ---------------------------------------------------------------------------
// tiled_v2 ("V2"): kernels.metal:462-596. K/V tiles of 16 keys double-buffered in
// LDS as fp32, filled with 128-bit global loads that are issued before the
// tile's arithmetic and written to the other buffer after it (one barrier per tile).
// ---------------------------------------------------------------------------
template <typename T> struct Vec16B;  // 16-byte global chunk -> floats
template <> struct Vec16B<float> {
  static constexpr int N = 4;
  __device__ static void load(const float *p, bool in, float (&out)[4]) {
    float4 v = in ? *reinterpret_cast<const float4 *>(p) : make_float4(0, 0, 0, 0);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
};
template <> struct Vec16B<_Float16> {
  static constexpr int N = 8;
  __device__ static void load(const _Float16 *p, bool in, float (&out)[8]) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in) v = *reinterpret_cast<const h8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
  }
};
template <> struct Vec16B<__bf16> {
  static constexpr int N = 8;
  __device__ static void load(const __bf16 *p, bool in, float (&out)[8]) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    b8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in) v = *reinterpret_cast<const b8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
  }
};

// A query row is shared by FOUR lanes (each owns D/4 head-dim elements of q and of the accumulator; a score is the sum
// of four partial dot products, two quad-permute DPP adds), a lane serves TWO rows (every K/V value read from LDS feeds
// both), a wave therefore holds 32 rows, and the EIGHT waves of a workgroup split the keys of those 32 rows (wave w
// takes the 16-key tiles t = w, w+8, ... -- strided, so causal blocks stay balanced), each with its own double-buffered
// fp32 K/V tiles; one merge through LDS by the row maxima at the end.
// One thread per row (the reference's shape, kernels.metal:462-596, and round 2's) put BASELINE config 2 -- 8 heads x
// 1024 rows -- on 128 waves of a 1024-SIMD chip: 676 us. This shape gives it 2048 waves.
constexpr int V2_BR = 32, V2_NW = 8, V2_RPL = 2;
template <int D> constexpr int v2_bc() { return (D <= 64) ? 16 : 8; }  // keys per tile (128 KiB of LDS per workgroup at D >= 64)
template <int D> constexpr size_t v2_lds_bytes() {  // the tile buffers, reused by the merge ([NW][RPL * (D/4 + 2)][64] floats)
  constexpr size_t tiles = (size_t)2 * V2_NW * 2 * v2_bc<D>() * D, merge = (size_t)V2_NW * V2_RPL * (D / 4 + 2) * 64;
  return (tiles > merge ? tiles : merge) * sizeof(float);
}

template <typename T, int D>
__global__ __launch_bounds__(64 * V2_NW, 1) void tiled_v2_kernel(Params p) {
  constexpr int BR = V2_BR, NW = V2_NW, RPL = V2_RPL;
  constexpr int BC = v2_bc<D>();
  constexpr int DS = D / 4;                    // head-dim elements per lane
  constexpr int EPC = Vec16B<T>::N;            // elements per 16-byte chunk
  constexpr int CHUNKS = BC * D / EPC;         // chunks per tile (K or V)
  constexpr int CPT = (CHUNKS + 63) / 64;      // chunks per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];  // K tiles, then V tiles: [NW][2][BC * D] each
  auto Ks = [&](int wv, int buf) { return smem + (wv * 2 + buf) * (BC * D); };
  auto Vs = [&](int wv, int buf) { return smem + ((NW + wv) * 2 + buf) * (BC * D); };

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int rg = lane >> 2, sl = lane & 3;
  const int row0 = blockIdx.x * BR;
  int row[RPL];
  bool valid[RPL];
  const long long base = (long long)blockIdx.z * p.batch_stride + (long long)blockIdx.y * p.head_stride;
  const T *Q = (const T *)p.q + base, *K = (const T *)p.k + base, *V = (const T *)p.v + base;
  T *O = (T *)p.o + base;

  float qreg[RPL][DS], acc[RPL][DS], m[RPL], l[RPL];
#pragma unroll
  for (int x = 0; x < RPL; ++x) {
    row[x] = row0 + 16 * x + rg;
    valid[x] = row[x] < p.N;
#pragma unroll
    for (int c = 0; c < DS / EPC; ++c) {
      float tmp[EPC];
      Vec16B<T>::load(Q + (long long)row[x] * D + sl * DS + c * EPC, valid[x], tmp);
#pragma unroll
      for (int e = 0; e < EPC; ++e) qreg[x][c * EPC + e] = tmp[e];
    }
#pragma unroll
    for (int d = 0; d < DS; ++d) acc[x][d] = 0.0f;
    m[x] = -INFINITY;
    l[x] = 0.0f;
  }

  const int last_row = min(row0 + BR, p.N) - 1;
  const int kv_end = p.is_causal ? last_row + 1 : p.N;
  const int ntiles = (kv_end + BC - 1) / BC;
  const int steps = (ntiles + NW - 1) / NW;    // every wave runs the same number of steps (one barrier each)

  float kst[CPT][EPC], vst[CPT][EPC];
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = lane + i * 64;
      const int e0 = c * EPC;
      const int r = e0 / D;
      const bool in = (c < CHUNKS) && (t < ntiles) && (t * BC + r < p.N);
      const long long g = (long long)(t * BC) * D + e0;
      Vec16B<T>::load(K + g, in, kst[i]);
      Vec16B<T>::load(V + g, in, vst[i]);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = lane + i * 64;
      if (c < CHUNKS) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          *reinterpret_cast<float4 *>(Ks(w, buf) + c * EPC + e) =
              make_float4(kst[i][e], kst[i][e + 1], kst[i][e + 2], kst[i][e + 3]);
          *reinterpret_cast<float4 *>(Vs(w, buf) + c * EPC + e) =
              make_float4(vst[i][e], vst[i][e + 1], vst[i][e + 2], vst[i][e + 3]);
        }
      }
    }
  };
  // sum over the four lanes of a row: quad_perm [1,0,3,2] then [2,3,0,1]
  auto quad_sum = [](float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
  };

  issue(w);
  commit(0);
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int t = st * NW + w;
    const int buf = st & 1;
    if (st + 1 < steps) issue(t + NW);  // in flight under this tile's arithmetic
    if (t < ntiles) {
      const int kv0 = t * BC;
      float s[RPL][BC];
      float tmax[RPL];
#pragma unroll
      for (int x = 0; x < RPL; ++x) tmax[x] = -INFINITY;
#pragma unroll
      for (int j = 0; j < BC; ++j) {
        const float4 *kr = reinterpret_cast<const float4 *>(Ks(w, buf) + j * D + sl * DS);
        float a[RPL];
#pragma unroll
        for (int x = 0; x < RPL; ++x) a[x] = 0.0f;
#pragma unroll
        for (int c = 0; c < DS / 4; ++c) {
          const float4 kk = kr[c];
#pragma unroll
          for (int x = 0; x < RPL; ++x)  // explicit FMAs: the file is built with -ffp-contract=off (a mul + an add per MAC otherwise)
            a[x] = __builtin_fmaf(qreg[x][4 * c + 3], kk.w, __builtin_fmaf(qreg[x][4 * c + 2], kk.z,
                   __builtin_fmaf(qreg[x][4 * c + 1], kk.y, __builtin_fmaf(qreg[x][4 * c], kk.x, a[x]))));
        }
        const int key = kv0 + j;
#pragma unroll
        for (int x = 0; x < RPL; ++x) {
          const float sc = quad_sum(a[x]) * p.scale;
          const bool vis = key < p.N && (!p.is_causal || key <= row[x]);
          s[x][j] = vis ? sc : -INFINITY;
          tmax[x] = fmaxf(tmax[x], s[x][j]);
        }
      }
      // per-tile online softmax; a fully masked tile leaves a row's state untouched (its weights below are all 0)
      float pj[RPL][BC];
#pragma unroll
      for (int x = 0; x < RPL; ++x) {
        const float m_new = fmaxf(m[x], tmax[x]);
        const bool any = m_new != -INFINITY;
        // exp(x) = 2^(x log2 e) on the hardware exponential (1 ulp; expf's range handling costs ~10 instructions per call)
        const float alpha = any ? __builtin_amdgcn_exp2f((m[x] - m_new) * 1.4426950408889634f) : 1.0f;
        float psum = 0.0f;
#pragma unroll
        for (int d = 0; d < DS; ++d) acc[x][d] *= alpha;
#pragma unroll
        for (int j = 0; j < BC; ++j) {
          pj[x][j] = any ? __builtin_amdgcn_exp2f((s[x][j] - m_new) * 1.4426950408889634f) : 0.0f;
          psum += pj[x][j];
        }
        l[x] = l[x] * alpha + psum;
        m[x] = m_new;
      }
#pragma unroll
      for (int j = 0; j < BC; ++j) {
        const float4 *vr = reinterpret_cast<const float4 *>(Vs(w, buf) + j * D + sl * DS);
#pragma unroll
        for (int c = 0; c < DS / 4; ++c) {
          const float4 vv = vr[c];
#pragma unroll
          for (int x = 0; x < RPL; ++x) {
            acc[x][4 * c] = __builtin_fmaf(pj[x][j], vv.x, acc[x][4 * c]);
            acc[x][4 * c + 1] = __builtin_fmaf(pj[x][j], vv.y, acc[x][4 * c + 1]);
            acc[x][4 * c + 2] = __builtin_fmaf(pj[x][j], vv.z, acc[x][4 * c + 2]);
            acc[x][4 * c + 3] = __builtin_fmaf(pj[x][j], vv.w, acc[x][4 * c + 3]);
          }
        }
      }
    }
    if (st + 1 < steps) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- merge the key splits (a wave that saw no visible key of a row has m = -inf, l = 0: weight 0)
  constexpr int MW = RPL * (DS + 2);  // floats per (wave, lane)
  float *mb = smem;                   // [NW][MW][64] floats inside the tile buffers (all reads are done: last barrier)
  static_assert((size_t)NW * MW * 64 * sizeof(float) <= v2_lds_bytes<D>(), "merge buffer fits the LDS allocation");
#pragma unroll
  for (int x = 0; x < RPL; ++x) {
#pragma unroll
    for (int d = 0; d < DS; ++d) mb[(w * MW + x * (DS + 2) + d) * 64 + lane] = acc[x][d];
    mb[(w * MW + x * (DS + 2) + DS) * 64 + lane] = m[x];
    mb[(w * MW + x * (DS + 2) + DS + 1) * 64 + lane] = l[x];
  }
  __syncthreads();
  if (w >= RPL) return;  // wave x finishes row set x
  const int x = w;
  const int orow = row0 + 16 * x + rg;
  if (orow >= p.N) return;
  float M = -INFINITY;
#pragma unroll
  for (int u = 0; u < NW; ++u) M = fmaxf(M, mb[(u * MW + x * (DS + 2) + DS) * 64 + lane]);
  float L = 0.0f, out[DS];
#pragma unroll
  for (int d = 0; d < DS; ++d) out[d] = 0.0f;
#pragma unroll
  for (int u = 0; u < NW; ++u) {
    const float mu = mb[(u * MW + x * (DS + 2) + DS) * 64 + lane];
    const float wu = (mu == -INFINITY) ? 0.0f : expf(mu - M);
    L += wu * mb[(u * MW + x * (DS + 2) + DS + 1) * 64 + lane];
#pragma unroll
    for (int d = 0; d < DS; ++d) out[d] += wu * mb[(u * MW + x * (DS + 2) + d) * 64 + lane];
  }
  const float inv = 1.0f / L;
#pragma unroll
  for (int d = 0; d < DS; ++d) st_elem(O, (long long)orow * D + sl * DS + d, out[d] * inv);
  if (p.lse && sl == 0) {
    const long long bh = (long long)blockIdx.z * p.H + blockIdx.y;
    p.lse[bh * p.N + orow] = M + logf(L);
  }
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
static bool d_ok(int D) { return D == 32 || D == 64 || D == 128; }
bool naive_supported(int dtype, int D) { return dtype >= FA_DTYPE_F32 && dtype <= FA_DTYPE_BF16 && d_ok(D); }
bool tiled_supported(int dtype, int D) { return naive_supported(dtype, D); }
bool tiled_v2_supported(int dtype, int D) { return naive_supported(dtype, D); }

#define FA_DISPATCH_TD(KERNEL, grid, block)                                          \
  do {                                                                               \
    switch (dtype) {                                                                 \
      case FA_DTYPE_F32:                                                             \
        if (p.D == 32) hipLaunchKernelGGL((KERNEL<float, 32>), grid, block, 0, s, p);       \
        else if (p.D == 64) hipLaunchKernelGGL((KERNEL<float, 64>), grid, block, 0, s, p);  \
        else hipLaunchKernelGGL((KERNEL<float, 128>), grid, block, 0, s, p);                \
        break;                                                                       \
      case FA_DTYPE_F16:                                                             \
        if (p.D == 32) hipLaunchKernelGGL((KERNEL<_Float16, 32>), grid, block, 0, s, p);    \
        else if (p.D == 64) hipLaunchKernelGGL((KERNEL<_Float16, 64>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((KERNEL<_Float16, 128>), grid, block, 0, s, p);             \
        break;                                                                       \
      default:                                                                       \
        if (p.D == 32) hipLaunchKernelGGL((KERNEL<__bf16, 32>), grid, block, 0, s, p);      \
        else if (p.D == 64) hipLaunchKernelGGL((KERNEL<__bf16, 64>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((KERNEL<__bf16, 128>), grid, block, 0, s, p);               \
        break;                                                                       \
    }                                                                                \
  } while (0)

hipError_t launch_naive(const Params &p, int dtype, hipStream_t s) {
  dim3 grid((p.N + 255) / 256, p.H, p.B), block(256);
  FA_DISPATCH_TD(naive_kernel, grid, block);
  return hipGetLastError();
}
hipError_t launch_tiled(const Params &p, int dtype, hipStream_t s) {
  dim3 grid((p.N + 63) / 64, p.H, p.B), block(64);
  FA_DISPATCH_TD(tiled_kernel, grid, block);
  return hipGetLastError();
}
template <typename T, int D>
static hipError_t launch_tiled_v2_one(const Params &p, hipStream_t s) {
  auto kern = tiled_v2_kernel<T, D>;
  // once per (kernel, device), marked done only on success (as the matrix-core launchers do)
  const hipError_t attr = set_dyn_lds_once((const void *)kern, (int)v2_lds_bytes<D>());
  if (attr != hipSuccess) return attr;
  dim3 grid((p.N + V2_BR - 1) / V2_BR, p.H, p.B), block(64 * V2_NW);
  (void)hipGetLastError();
  hipLaunchKernelGGL(kern, grid, block, v2_lds_bytes<D>(), s, p);
  return hipGetLastError();
}
template <typename T>
static hipError_t launch_tiled_v2_t(const Params &p, hipStream_t s) {
  if (p.D == 32) return launch_tiled_v2_one<T, 32>(p, s);
  if (p.D == 64) return launch_tiled_v2_one<T, 64>(p, s);
  return launch_tiled_v2_one<T, 128>(p, s);
}
hipError_t launch_tiled_v2(const Params &p, int dtype, hipStream_t s) {
  if (dtype == FA_DTYPE_F32) return launch_tiled_v2_t<float>(p, s);
  return dtype == FA_DTYPE_F16 ? launch_tiled_v2_t<_Float16>(p, s) : launch_tiled_v2_t<__bf16>(p, s);
}

}  // namespace fa
