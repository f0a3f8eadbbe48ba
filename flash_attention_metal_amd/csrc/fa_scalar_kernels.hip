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
// tiled_v2 ("V2"): kernels.metal:462-596. 128 rows per block (2 waves),
// Bc = 16, K/V tiles double-buffered in LDS as fp32, filled with 128-bit
// global loads that are issued before the tile's arithmetic and written to the
// other buffer after it (one barrier per tile).
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

template <typename T, int D>
__global__ __launch_bounds__(128) void tiled_v2_kernel(Params p) {
  constexpr int BR = 128, BC = 16;
  constexpr int EPC = Vec16B<T>::N;            // elements per 16-byte chunk
  constexpr int CHUNKS = BC * D / EPC;         // chunks per tile (K or V)
  constexpr int CPT = (CHUNKS + BR - 1) / BR;  // chunks per thread
  __shared__ __attribute__((aligned(16))) float Ks[2][BC * D];
  __shared__ __attribute__((aligned(16))) float Vs[2][BC * D];

  const int tx = threadIdx.x;
  const int row0 = blockIdx.x * BR;
  const int row = row0 + tx;
  const bool valid = row < p.N;
  const long long base = (long long)blockIdx.z * p.batch_stride + (long long)blockIdx.y * p.head_stride;
  const T *Q = (const T *)p.q + base, *K = (const T *)p.k + base, *V = (const T *)p.v + base;
  T *O = (T *)p.o + base;

  float qreg[D], acc[D];
#pragma unroll
  for (int c = 0; c < D / EPC; ++c) {
    float tmp[EPC];
    Vec16B<T>::load(Q + (long long)row * D + c * EPC, valid, tmp);
#pragma unroll
    for (int e = 0; e < EPC; ++e) qreg[c * EPC + e] = tmp[e];
  }
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.0f;
  float m = -INFINITY, l = 0.0f;

  const int last_row = min(row0 + BR, p.N) - 1;
  const int kv_end = p.is_causal ? last_row + 1 : p.N;
  const int ntiles = (kv_end + BC - 1) / BC;

  float kst[CPT][EPC], vst[CPT][EPC];
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tx + i * BR;
      const int e0 = c * EPC;
      const int r = e0 / D;
      const bool in = (c < CHUNKS) && (t * BC + r < p.N);
      const long long g = (long long)(t * BC) * D + e0;
      Vec16B<T>::load(K + g, in, kst[i]);
      Vec16B<T>::load(V + g, in, vst[i]);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tx + i * BR;
      if (c < CHUNKS) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          *reinterpret_cast<float4 *>(&Ks[buf][c * EPC + e]) =
              make_float4(kst[i][e], kst[i][e + 1], kst[i][e + 2], kst[i][e + 3]);
          *reinterpret_cast<float4 *>(&Vs[buf][c * EPC + e]) =
              make_float4(vst[i][e], vst[i][e + 1], vst[i][e + 2], vst[i][e + 3]);
        }
      }
    }
  };

  issue(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue(t + 1);  // in flight under this tile's arithmetic
    const int kv0 = t * BC;
    float s[BC];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < BC; ++j) {
      const float4 *kr = reinterpret_cast<const float4 *>(&Ks[buf][j * D]);
      float a = 0.0f;
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        const float4 kk = kr[c];
        a += qreg[4 * c] * kk.x + qreg[4 * c + 1] * kk.y + qreg[4 * c + 2] * kk.z + qreg[4 * c + 3] * kk.w;
      }
      a *= p.scale;
      const int key = kv0 + j;
      const bool vis = key < p.N && (!p.is_causal || key <= row);
      s[j] = vis ? a : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    // per-tile online softmax; a fully masked tile leaves the state untouched
    const float m_new = fmaxf(m, tmax);
    if (m_new != -INFINITY) {
      const float alpha = expf(m - m_new);
      float psum = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] *= alpha;
#pragma unroll
      for (int j = 0; j < BC; ++j) {
        const float pj = expf(s[j] - m_new);
        psum += pj;
        const float4 *vr = reinterpret_cast<const float4 *>(&Vs[buf][j * D]);
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
          const float4 vv = vr[c];
          acc[4 * c] += pj * vv.x;
          acc[4 * c + 1] += pj * vv.y;
          acc[4 * c + 2] += pj * vv.z;
          acc[4 * c + 3] += pj * vv.w;
        }
      }
      l = l * alpha + psum;
      m = m_new;
    }
    if (t + 1 < ntiles) commit(buf ^ 1);
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
hipError_t launch_tiled_v2(const Params &p, int dtype, hipStream_t s) {
  dim3 grid((p.N + 127) / 128, p.H, p.B), block(128);
  FA_DISPATCH_TD(tiled_v2_kernel, grid, block);
  return hipGetLastError();
}

}  // namespace fa
