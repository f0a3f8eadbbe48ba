// fa_mfma_common.h -- types and lane-level helpers shared by the matrix-core kernels.
#pragma once
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

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
// mfma_v0 / mfma_v: the same instruction through inline asm with the accumulator tuple forced into
// ARCHITECTURAL VGPRs ("v"). In a kernel that may use the accumulation file (one wave per SIMD,
// launch_bounds(256,1)) hipcc selects the AGPR form for every builtin MFMA; a score tile that the
// softmax reads next would then cost one v_accvgpr_read per element. hipcc pads no hazards around an
// asm statement: callers keep >= 12 wait states between the last mfma_v of a chain and the first VALU
// read of its result (cdna_hip_programming.md section 5.7 item 2).
#define FA_MFMA_ASM(OP)                                                                              \
  __device__ static __forceinline__ void mfma_v0(f32x16 &d, u32x4 a, u32x4 b) {                          \
    asm volatile(OP " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));                                  \
  }                                                                                                  \
  __device__ static __forceinline__ void mfma_v(f32x16 &d, u32x4 a, u32x4 b) {                           \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));                                  \
  }

template <> struct MT<BF16> {
  using elem = __bf16;
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  FA_MFMA_ASM("v_mfma_f32_32x32x16_bf16")
};
template <> struct MT<F16> {
  using elem = _Float16;
  using vec8 = f16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  FA_MFMA_ASM("v_mfma_f32_32x32x16_f16")
};

template <> struct MT<FP8> {  // e4m3 inputs, bf16 P and O: this is the PV product's instruction (and the score product's in the
  using elem = __bf16;         // kernels that widen K to bf16 while staging). The 128-row kernel multiplies e4m3 by e4m3 on
                               // v_mfma_scale_f32_32x32x64_f8f6f4 for the scores (fa_mfma_kernel.hip), twice the bf16 rate
  using vec8 = bf16x8;
  __device__ static __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  FA_MFMA_ASM("v_mfma_f32_32x32x16_bf16")
};


// ---- asm-owned accumulation registers -------------------------------------------------------------
// The paired-block kernel keeps O^T in FIXED accumulation registers a[0 : NACC) that only the inline
// asm below names: hipcc neither allocates nor copies them. (Left to the register allocator, 16-wide
// accumulator tuples that live across the loop's rare branches were copied on the COMMON path: 32
// v_accvgpr_mov per 32x64 score tile.) Every statement lists the whole range as clobbered, which also
// makes the kernel descriptor reserve it. hipcc pads no hazards around asm: see the callers' s_nop.
#define FA_A64 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
#define FA_A96 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95"
#define FA_A112 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111"
#define FA_A128 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"
#define FA_A192 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"
#define FA_A224 "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223"

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// One asm statement with the clobber list that matches the kernel's owned range a[0 : NACC).
#define FA_ACC_ASM_(NACC, PRE, ...)                                           \
  do {                                                                        \
    if constexpr ((NACC) == 64) asm volatile(__VA_ARGS__ : PRE FA_A64);       \
    else if constexpr ((NACC) == 96) asm volatile(__VA_ARGS__ : PRE FA_A96);  \
    else if constexpr ((NACC) == 112) asm volatile(__VA_ARGS__ : PRE FA_A112); \
    else if constexpr ((NACC) == 128) asm volatile(__VA_ARGS__ : PRE FA_A128); \
    else if constexpr ((NACC) == 192) asm volatile(__VA_ARGS__ : PRE FA_A192); \
    else asm volatile(__VA_ARGS__ : PRE FA_A224);                             \
  } while (0)
#define FA_ACC_ASM(NACC, ...) FA_ACC_ASM_(NACC, , __VA_ARGS__)
#define FA_MEM_CLOBBER "memory",
#define FA_ACC_ASM_MEM(NACC, ...) FA_ACC_ASM_(NACC, FA_MEM_CLOBBER, __VA_ARGS__)

template <typename Tag> struct MfmaOp { static constexpr bool is_f16 = std::is_same<Tag, F16>::value; };

// a[16*TI .. 16*TI+15] += A(frag) * B(frag)   (32x32x16, A/B in VGPRs)
template <typename Tag, int NACC, int TI>
__device__ __forceinline__ void acc_mfma(u32x4 a, u32x4 b) {
  constexpr int lo = 16 * TI, hi = lo + 15;
  static_assert(hi < NACC, "accumulator tuple out of range");
  if constexpr (MfmaOp<Tag>::is_f16) FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(lo), "i"(hi));
  else FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(lo), "i"(hi));
}
// d (VGPR tuple) = A(frag, VGPR) * B(a[QR .. QR+3], asm-owned) [+ d]: the score product with Q parked in the accumulation file
template <typename Tag, int NACC, int QR, bool FIRST>
__device__ __forceinline__ void mfma_v_qacc(f32x16 &d, u32x4 a) {
  static_assert(QR + 3 < NACC, "Q fragment out of range");
  if constexpr (FIRST) {
    if constexpr (MfmaOp<Tag>::is_f16) FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(QR), "i"(QR + 3));
    else FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(QR), "i"(QR + 3));
  } else {
    if constexpr (MfmaOp<Tag>::is_f16) FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(QR), "i"(QR + 3));
    else FA_ACC_ASM(NACC, "v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(QR), "i"(QR + 3));
  }
}
// global -> a[R..R+3] (16 bytes per lane) through a buffer descriptor. hipcc does not count an asm load: the
// caller waits with acc_lds_write_b128's vmcnt (loads complete in issue order).
template <int NACC, int R>
__device__ __forceinline__ void acc_buffer_load_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  static_assert(R + 3 < NACC, "staging registers out of range");
  FA_ACC_ASM(NACC, "buffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen" ::"v"(voff), "s"(rsrc), "s"(soff), "i"(R), "i"(R + 3));
}
// wait until at most PENDING younger vector-memory operations are outstanding, then LDS[addr + OFF] = a[R..R+3]
template <int NACC, int R, int PENDING, int OFF>
__device__ __forceinline__ void acc_lds_write_b128(unsigned lds_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  FA_ACC_ASM_MEM(NACC, "s_waitcnt vmcnt(%c3)\n\tds_write_b128 %0, a[%c1:%c2] offset:%c4" ::"v"(lds_addr), "i"(R), "i"(R + 3), "i"(PENDING), "i"(OFF));
}
template <int NACC, int R>
__device__ __forceinline__ void acc_zero1() {
  FA_ACC_ASM(NACC, "v_accvgpr_write_b32 a%c0, 0" ::"i"(R));
}
template <int NACC, int R>
__device__ __forceinline__ void acc_write1(unsigned v) {
  FA_ACC_ASM(NACC, "v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "i"(R));
}
template <int NACC, int R>
__device__ __forceinline__ float acc_read1() {
  float f;
  FA_ACC_ASM(NACC, "v_accvgpr_read_b32 %0, a%c1" : "=v"(f) : "i"(R));
  return f;
}
template <int NACC, int R>
__device__ __forceinline__ void acc_scale1(float alpha) {  // a[R] *= alpha (per lane)
  float t;
  FA_ACC_ASM(NACC, "v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(t) : "v"(alpha), "i"(R));
}

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
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// block id -> (bh = batch*H + head, q block). Divisors come precomputed in Params (set_block_divisors).
template <bool CAUSAL>
__device__ __forceinline__ void map_block(int id, const Params &p, int &bh, int &qb) {
  const int BH = p.B * p.H, nQ = p.nq;
  if (CAUSAL) {
    // heaviest-first within groups of HG heads (HG = BH: across all heads). A group keeps HG/8
    // heads per XCD in flight, which bounds the K/V working set of that XCD's L2.
    const int HG = p.hg;
    const int g = (int)fdiv((unsigned)id, p.fd_per), rem = id - g * HG * nQ;
    const int rq = (int)fdiv((unsigned)rem, p.fd_hg);
    qb = nQ - 1 - rq;
    bh = g * HG + (rem - rq * HG);  // HG % 8 == 0  =>  id % 8 == bh % 8: the head keeps its XCD residue
    return;
  }
  const int full = BH & ~7;  // heads that can be dealt 8 at a time
  if (id < full * nQ) {
    const int xcd = id & 7, slot = id >> 3;
    const int sq = (int)fdiv((unsigned)slot, p.fd_nq);
    bh = sq * 8 + xcd;
    qb = slot - sq * nQ;
  } else {
    const int rem = id - full * nQ;
    const int sq = (int)fdiv((unsigned)rem, p.fd_nq);
    bh = full + sq;
    qb = rem - sq * nQ;
  }
}

// The same map with plain divisions (backward kernels: BwdParams carries no precomputed divisors; one global group)
template <bool CAUSAL>
__device__ __forceinline__ void map_block_div(int id, int BH, int nQ, int &bh, int &qb) {
  if (CAUSAL) {
    const int rq = id / BH;
    qb = nQ - 1 - rq;
    bh = id - rq * BH;
    return;
  }
  const int full = BH & ~7;
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

// element offsets of head bh in Q/O and in K/V (grouped-query heads: query head h reads key/value head h / (H / Hkv))
__device__ __forceinline__ void head_bases(int bh, const Params &p, long long &base, long long &base_kv) {
  const int b = (int)fdiv((unsigned)bh, p.fd_h), h = bh - b * p.H;
  base = (long long)b * p.batch_stride + (long long)h * p.head_stride;
  base_kv = (long long)b * p.kv_batch_stride + (long long)fdiv((unsigned)h, p.fd_gq) * p.kv_head_stride;
}

// ---- host-side launch helpers shared by the matrix-core kernel files ---------------------------

// Causal issue order: heaviest-first within groups of `head_group` heads (map_block above).
// One global group balances best. Measured on config 3 (1 MiB of K+V per head, 8 heads per XCD
// in flight): 279 MB fetched per launch vs 140 MB with 32-head groups (algorithmic reads 101 MB),
// but the global order is 3.6 % FASTER (interleaved A/B) -- the re-reads are served by the
// 256 MiB Infinity Cache, not HBM. Groups are therefore only used where they cost nothing:
// long sequences, sized so that one XCD's share of K+V stays under 8 MiB (floor: 16 heads).
inline int causal_head_group(const Params &p, int D, int elem_bytes) {
  const int BH = p.B * p.H;
  const double kv_bytes = 2.0 * p.Nk * D * elem_bytes;
  int per_xcd = (int)(8.0 * 1024 * 1024 / kv_bytes);
  if (per_xcd < 2) per_xcd = 2;
  int hg = 8 * per_xcd;
  while (hg < BH && (BH % hg) != 0) hg += 8;
  return (BH % 8 == 0 && hg < BH) ? hg : 0;
}

// Fill the divisors map_block / head_bases use: nq = q blocks per head at this kernel's block height, head_group as
// returned by causal_head_group (0 or anything that does not divide B*H = one group).
inline void set_block_divisors(Params &p, int nq, int head_group) {
  const int BH = p.B * p.H;
  int hg = head_group;
  if (hg <= 0 || hg > BH || BH % hg != 0) hg = BH;
  p.nq = nq;
  p.hg = hg;
  p.fd_h = make_fastdiv((unsigned)p.H);
  p.fd_gq = make_fastdiv((unsigned)(p.H / p.Hkv));
  p.fd_nq = make_fastdiv((unsigned)nq);
  p.fd_hg = make_fastdiv((unsigned)hg);
  p.fd_per = make_fastdiv((unsigned)hg * (unsigned)nq);
}

}  // namespace fa
