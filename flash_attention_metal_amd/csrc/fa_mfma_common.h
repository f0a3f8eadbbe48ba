// fa_mfma_common.h -- types and lane-level helpers shared by the matrix-core kernels.
#pragma once
#include <type_traits>

#include "fa_common.h"

namespace fa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4), may_alias));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2), may_alias));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename Tag> struct MT;
template <> struct MT<BF16> {
  using elem = __bf16;
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MT<F16> {
  using elem = _Float16;
  using vec8 = f16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <> struct MT<FP8> {  // e4m3 inputs: converted to bf16 on the way into registers / LDS (exact),
  using elem = __bf16;         // bf16 MFMA from there on (non-scaled fp8 MFMA has the same rate), O is bf16
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

constexpr int BM = 128;      // query rows per workgroup
constexpr int WM = 32;       // query rows per wave
constexpr int BN = 64;       // keys per tile
constexpr int NTHREADS = 256;

typedef __attribute__((address_space(3))) char lds_char;

// 8 OCP e4m3 values (two dwords) -> 8 bf16 (four dwords); exact: e4m3 is a subset of bf16
__device__ __forceinline__ u32x4 fp8x8_to_bf16(u32x2 w) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
    const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
    bf16x2 a, b;
    a[0] = (__bf16)lo[0]; a[1] = (__bf16)lo[1];
    b[0] = (__bf16)hi[0]; b[1] = (__bf16)hi[1];
    out[2 * i] = __builtin_bit_cast(unsigned, a);
    out[2 * i + 1] = __builtin_bit_cast(unsigned, b);
  }
  return out;
}

__device__ __forceinline__ u32x4 lds_read_b128(const lds_char *p) {
  return *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(p);
}
__device__ __forceinline__ void lds_write_b128(lds_char *p, u32x4 v) {
  *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(p) = v;
}
// 4-byte LDS store that may alias the u32x4 loads above (a `float` store next to a `u32x4` load
// of the same bytes is a strict-aliasing violation: hipcc then folded the 4 loaded lanes into one)
typedef unsigned int __attribute__((may_alias)) u32_alias;
__device__ __forceinline__ void lds_write_b32(lds_char *p, unsigned v) {
  *reinterpret_cast<__attribute__((address_space(3))) u32_alias *>(p) = v;
}
__device__ __forceinline__ void lds_write_b64(lds_char *p, u32x2 v) {
  *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(p) = v;
}
// transposed 4x16 block read (ds_read_b64_tr_b16): lane i of a 16-lane group
// receives column i of the 4 rows whose addresses lanes 4q+p supplied
__device__ __forceinline__ s16x4 lds_read_tr16(const lds_char *p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(p));
}

// Combine a per-lane value with the one held by lane^32 (the other half of the
// same query row). v_permlane32_swap exchanges vdst[32..63] with src[0..31], so
// with both operands holding x the pair becomes {x_lo|x_lo, x_hi|x_hi}
// (checked on hardware: tools/probe_layouts.hip).
// Inline asm on purpose: with __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2)
// used the FIRST result for both elements here (.s: v_add_f32 v2, v34, v34), so
// the halves never met. The s_nop covers the VALU-write -> permlane-read hazard
// (2 wait states), which hipcc does not pad inside an asm string.
__device__ __forceinline__ void half_pair(float x, float &lo, float &hi) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}

// block id -> (batch*head, q block). Blocks b and b+8 share an XCD's L2 (observed dispatch
// order; speed only, never correctness), so heads are dealt to the 8 residues: one head's
// K/V then stays in one L2.
//  * non-causal (uniform work): a head's q blocks are consecutive within its residue.
//  * causal (q block i costs i+1 tiles): blocks are issued heaviest-first ACROSS all heads --
//    q block nQ-1 of every head, then nQ-2 of every head, ... . Heavy-first only within a
//    group of heads left the last group's 64-tile blocks starting late: measured 28 % over
//    the ideal at N=4096 (simulated makespan 122 vs 88 tile-times; this order: 98).
template <bool CAUSAL>
__device__ __forceinline__ void map_block(int id, int BH, int nQ, int &bh, int &qb, int HG = 0) {
  if (CAUSAL) {
    // heaviest-first within groups of HG heads (HG = BH: across all heads). A group keeps HG/8
    // heads per XCD in flight, which bounds the K/V working set of that XCD's L2.
    if (HG <= 0 || HG > BH || BH % HG != 0) HG = BH;
    const int per = HG * nQ, g = id / per, rem = id - g * per;
    qb = nQ - 1 - rem / HG;
    bh = g * HG + rem % HG;  // HG % 8 == 0  =>  id % 8 == bh % 8: the head keeps its XCD residue
    return;
  }
  const int full = (BH / 8) * 8;  // heads that can be dealt 8 at a time
  if (id < full * nQ) {
    const int xcd = id & 7, slot = id >> 3;
    bh = (slot / nQ) * 8 + xcd;
    qb = slot % nQ;
  } else {
    const int rem = id - full * nQ;
    bh = full + rem / nQ;
    qb = rem % nQ;
  }
}

}  // namespace fa
