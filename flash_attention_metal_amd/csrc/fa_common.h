// fa_common.h -- internal declarations shared by the gfx950 kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/fa_mi355.h"

namespace fa {

// One launch of the operator (mirror of the binding table kernels.metal:600-613).
// Exact unsigned division by a run-time constant (Granlund-Montgomery): q = (t + ((n - t) >> sh1)) >> sh2 with
// t = mulhi(mul, n). The kernels map a block id to (batch, head, q block) with five divisions by launch constants; as
// generic divisions they cost ~30 instructions each and sat in front of the first global load of every workgroup.
struct FastDiv {
  unsigned mul = 1, sh1 = 0, sh2 = 0;  // default: divide by 1
};
inline FastDiv make_fastdiv(unsigned d) {  // d >= 1
  unsigned l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  FastDiv f;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1u;
  f.sh2 = l > 1 ? l - 1 : 0u;
  return f;
}

struct Params {
  const void *q, *k, *v;
  void *o;
  float *lse;  // nullable
  int B, H, N, D;
  float scale;
  long long batch_stride, head_stride;  // elements
  int is_causal;
  // generalised operator (fa_fwd_ex; fa_fwd sets Nk = N, Hkv = H, kv strides = q strides):
  int Nk = 0, Hkv = 0;                          // keys per head; key/value heads (H % Hkv == 0)
  long long kv_batch_stride = 0, kv_head_stride = 0;
  int exact = 0;       // 128-row kernel: 1 = no pre-scaled query operand (variant mfma_exact)
  int head_group = 0;  // internal: causal blocks are issued heaviest-first within groups of this many heads (0 = all)
  // internal, filled by the matrix-core launchers (set_block_divisors): divisors of the block id -> (batch, head, q block) map
  int nq = 0, hg = 0;  // q blocks per head for this kernel's block height; effective head group (a divisor of B*H)
  FastDiv fd_h, fd_gq, fd_nq, fd_hg, fd_per;  // by H, H/Hkv, nq, hg, hg*nq
};

// one fa_fwd_decode call (csrc/fa_decode_kernel.hip)
struct DecodeParams {
  const void *q, *k, *v;
  void *o;
  float *lse;  // nullable
  float *ws;   // workspace: per item [16 QT rows][D] partial O (unnormalised), then [16 QT rows][2] (m, l)
  int B, Hq, Hkv, Nq, Nk;
  float scale;
  long long q_bs, q_hs, kv_bs, kv_hs;  // elements
  int is_causal;
  int S;  // key splits per (batch, key head)
  int q8 = 0;  // e4m3 K / V: the queries are e4m3 too (else 16-bit)
};

// dtype tags
struct F32 {};
struct F16 {};
struct BF16 {};
struct FP8 {};

// launchers: return hipError_t of the launch
hipError_t launch_naive(const Params &p, int dtype, hipStream_t s);
hipError_t launch_tiled(const Params &p, int dtype, hipStream_t s);
hipError_t launch_tiled_v2(const Params &p, int dtype, hipStream_t s);
hipError_t launch_mfma(const Params &p, int dtype, hipStream_t s);
hipError_t launch_mfma_split2(const Params &p, int dtype, hipStream_t s);
bool mfma_split2_supported(int dtype, int D);
hipError_t launch_mfma_h64s2(const Params &p, int dtype, hipStream_t s);
bool mfma_h64s2_supported(int dtype, int D);
hipError_t launch_mfma16(const Params &p, int dtype, hipStream_t s);
bool mfma16_supported(int dtype, int D);
int mfma16_waves(int D, int BH, int N, int Nk, int is_causal);  // 4 or 8 waves (128 / 256 query rows) per workgroup
hipError_t launch_fp8pv(const Params &p, int dtype, hipStream_t s);
bool fp8pv_supported(int dtype, int D);
hipError_t launch_splitkv(const Params &p, int dtype, hipStream_t s);

hipError_t launch_decode(const DecodeParams &p, int D, int dtype, int kv8, hipStream_t s);
bool decode_supported(int dtype, int D);
int decode_splits(int B, int Hkv, int Nk, int D, int kv8);  // kv8: e4m3 inputs (one LDS image per item: twice the items per CU)
long long decode_workspace_bytes(int B, int Hq, int Hkv, int Nq, int Nk, int D);
bool naive_supported(int dtype, int D);
bool tiled_supported(int dtype, int D);
bool tiled_v2_supported(int dtype, int D);
bool mfma_supported(int dtype, int D);
bool splitkv_supported(int dtype, int D);
int splitkv_waves(int D, int Nk);
bool bwd_supported(int dtype, int D);
hipError_t launch_widen_e4m3(const void *in, void *out, long long n, hipStream_t s);  // e4m3 -> bf16, n % 16 == 0
hipError_t launch_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse,
                      float *dq, float *dk, float *dv, float *ws, int B, int H, int Hkv, int N, int Nk, int D, float scale,
                      long long bs, long long hs, long long kv_bs, long long kv_hs, int causal, int dtype, hipStream_t s);

// ---- host-side launch helper shared by every kernel file that needs more than 48 KiB of dynamic LDS
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch.
inline hipError_t set_dyn_lds_once(const void *fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void *, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(mu);
  if (done.count({fn, dev})) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({fn, dev});
  return e;
}

// ---- element load/store helpers for the scalar kernels -------------------
__device__ __forceinline__ float ld_elem(const float *p, long long i) { return p[i]; }
__device__ __forceinline__ float ld_elem(const _Float16 *p, long long i) { return (float)p[i]; }
__device__ __forceinline__ float ld_elem(const __bf16 *p, long long i) { return (float)p[i]; }
__device__ __forceinline__ void st_elem(float *p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void st_elem(_Float16 *p, long long i, float v) { p[i] = (_Float16)v; }
__device__ __forceinline__ void st_elem(__bf16 *p, long long i, float v) { p[i] = (__bf16)v; }

}  // namespace fa
