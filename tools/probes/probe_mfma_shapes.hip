// probe_mfma_shapes.hip -- throughput of an attention-like inner loop (LDS fragment reads, score MFMAs, exp2 + row sum +
// pack, PV MFMAs; junk math, real instruction mix and random operands) built on v_mfma_f32_32x32x16_bf16 versus
// v_mfma_f32_16x16x32_bf16: same FLOPs, same LDS bytes, same VALU per tile; three waves per SIMD like the 128-row kernel.
// The question: the forward is power-limited on random data, and MI355X_MICROARCH.md (DVFS
// give-back, item 7) reports the 16x16x32 shape at 1.12-1.15x the FLOP/s of 32x32x16 in bare loops. Does that survive the
// doubled MFMA issue slots once the softmax's VALU shares the port?
// build: hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize probe_mfma_shapes.hip -o probe_mfma_shapes ; run: ./probe_mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4), may_alias));
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ bf16x8 ldsr(const lds_char *p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(p));
}

template <int SHAPE, bool VALU>
__global__ __launch_bounds__(256, 3) void probe(const unsigned *in, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  lds_char *smem = (lds_char *)smem_g;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 4; i += 256) ((__attribute__((address_space(3))) unsigned *)smem)[i] = in[(blockIdx.x * 8192 + i) & 0xfffff];
  __syncthreads();
  bf16x8 qf[4];
  for (int i = 0; i < 4; ++i) {
    u32x4 t;
    for (int j = 0; j < 4; ++j) t[j] = in[(tid * 16 + i * 4 + j) & 0xfffff];
    qf[i] = __builtin_bit_cast(bf16x8, t);
  }
  // conflict-free fragment addresses: 128-byte rows, 16-byte chunk index XOR (row >> 1) & 7 (the kernels' K image)
  const int r32 = lane & 31, h32 = lane >> 5, m16 = lane & 15, g16 = lane >> 4;
  int a32[4], a16[2];
  for (int ks = 0; ks < 4; ++ks) a32[ks] = r32 * 128 + (((2 * ks + h32) ^ ((r32 >> 1) & 7)) << 4);
  for (int ks = 0; ks < 2; ++ks) a16[ks] = m16 * 128 + (((4 * ks + g16) ^ ((m16 >> 1) & 7)) << 4);
  const lds_char *base = smem;
  float l = 0.0f;
  if constexpr (SHAPE == 32) {
    f32x16 o[2], negm;
    for (int i = 0; i < 16; ++i) { o[0][i] = 0; o[1][i] = 0; negm[i] = -1.0f; }
    asm volatile("" : "+v"(negm));
    for (int it = 0; it < iters; ++it) {
      f32x16 s[2];
      const lds_char *b = base + (it & 1) * 16384;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr(b + kb * 4096 + a32[ks]), qf[ks], ks == 0 ? negm : s[kb], 0, 0, 0);
      bf16x8 pf[4];
      if constexpr (VALU) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) { s[kb][i] = __builtin_amdgcn_exp2f(s[kb][i]); l += s[kb][i]; }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[2 * kb + st][j] = (__bf16)s[kb][8 * st + j];
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr(b + 8192 + (f >> 1) * 4096 + a32[2 * (f & 1) + db]), pf[f], o[db], 0, 0, 0);
    }
    float acc = l;
    for (int i = 0; i < 16; ++i) acc += o[0][i] + o[1][i];
    out[blockIdx.x * 256 + tid] = acc;
  } else {
    f32x4 o[4][2], negm;
    for (int d = 0; d < 4; ++d) for (int q = 0; q < 2; ++q) for (int i = 0; i < 4; ++i) o[d][q][i] = 0;
    for (int i = 0; i < 4; ++i) negm[i] = -1.0f;
    asm volatile("" : "+v"(negm));
    for (int it = 0; it < iters; ++it) {
      f32x4 s[4][2];
      const lds_char *b = base + (it & 1) * 16384;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 a = ldsr(b + kt * 2048 + a16[ks]);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[2 * qt + ks], ks == 0 ? negm : s[kt][qt], 0, 0, 0);
        }
      bf16x8 pf[2][2];
      if constexpr (VALU) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[kt][qt][i] = __builtin_amdgcn_exp2f(s[kt][qt][i]); l += s[kt][qt][i]; }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qt][kp][j] = (__bf16)s[2 * kp + (j >> 2)][qt][j & 3];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          const bf16x8 v = ldsr(b + 8192 + dt * 2048 + a16[kp]);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, pf[qt][kp], o[dt][qt], 0, 0, 0);
        }
    }
    float acc = l;
    for (int d = 0; d < 4; ++d) for (int q = 0; q < 2; ++q) for (int i = 0; i < 4; ++i) acc += o[d][q][i];
    out[blockIdx.x * 256 + tid] = acc;
  }
}

// 32x32x16, 64 query rows per wave: every K / V fragment read from LDS feeds TWO MFMAs (half the LDS bytes per FLOP)
template <bool VALU>
__global__ __launch_bounds__(256, 2) void probe2(const unsigned *in, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  lds_char *smem = (lds_char *)smem_g;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 4; i += 256) ((__attribute__((address_space(3))) unsigned *)smem)[i] = in[(blockIdx.x * 8192 + i) & 0xfffff];
  __syncthreads();
  bf16x8 qf[2][4];
  for (int x = 0; x < 2; ++x)
    for (int i = 0; i < 4; ++i) {
      u32x4 t;
      for (int j = 0; j < 4; ++j) t[j] = in[(tid * 32 + x * 16 + i * 4 + j) & 0xfffff];
      qf[x][i] = __builtin_bit_cast(bf16x8, t);
    }
  const int r32 = lane & 31, h32 = lane >> 5;
  int a32[4];
  for (int ks = 0; ks < 4; ++ks) a32[ks] = r32 * 128 + (((2 * ks + h32) ^ ((r32 >> 1) & 7)) << 4);
  float l = 0.0f;
  f32x16 o[2][2], negm[2];
  for (int x = 0; x < 2; ++x) for (int i = 0; i < 16; ++i) { o[x][0][i] = 0; o[x][1][i] = 0; negm[x][i] = -1.0f; }
  asm volatile("" : "+v"(negm[0]), "+v"(negm[1]));
  for (int it = 0; it < iters; ++it) {
    f32x16 s[2][2];
    const lds_char *b = smem + (it & 1) * 16384;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = ldsr(b + kb * 4096 + a32[ks]);
#pragma unroll
        for (int x = 0; x < 2; ++x) s[x][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[x][ks], ks == 0 ? negm[x] : s[x][kb], 0, 0, 0);
      }
    bf16x8 pf[2][4];
    if constexpr (VALU) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) { s[x][kb][i] = __builtin_amdgcn_exp2f(s[x][kb][i]); l += s[x][kb][i]; }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[x][2 * kb + st][j] = (__bf16)s[x][kb][8 * st + j];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8 v = ldsr(b + 8192 + (f >> 1) * 4096 + a32[2 * (f & 1) + db]);
#pragma unroll
        for (int x = 0; x < 2; ++x) o[x][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, pf[x][f], o[x][db], 0, 0, 0);
      }
  }
  float acc = l;
  for (int x = 0; x < 2; ++x) for (int i = 0; i < 16; ++i) acc += o[x][0][i] + o[x][1][i];
  out[blockIdx.x * 256 + tid] = acc;
}

template <bool VALU>
static double run2(const unsigned *din, float *dout, int grid, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((probe2<VALU>), dim3(grid), dim3(256), 32768, 0, din, dout, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((probe2<VALU>), dim3(grid), dim3(256), 32768, 0, din, dout, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return (double)grid * 4 * iters * 32 * 32768.0 * 10 / (ms * 1e-3) / 1e12;
}

template <int SHAPE, bool VALU>
static double run(const unsigned *din, float *dout, int grid, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((probe<SHAPE, VALU>), dim3(grid), dim3(256), 32768, 0, din, dout, iters);  // ~0.4 s of warm-up
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((probe<SHAPE, VALU>), dim3(grid), dim3(256), 32768, 0, din, dout, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return (double)grid * 4 * iters * 16 * 32768.0 * 10 / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
  const int zeros = argc > 1 && atoi(argv[1]);
  std::vector<unsigned> h(1 << 20);
  srand(1);
  for (auto &x : h) {  // two bf16 in U(-1,1): sign, exponent 120..126, random mantissa
    auto one = [] { return (unsigned)(((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127)); };
    x = zeros ? 0u : (one() | (one() << 16));
  }
  unsigned *din; float *dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 768 * 256 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int grid = 768, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    printf("%s data: 32x32x16 + softmax VALU %7.1f TF | 16x16x32 + softmax VALU %7.1f TF | 32x32x16 bare %7.1f TF | 16x16x32 bare %7.1f TF\n",
           zeros ? "zero  " : "random", run<32, true>(din, dout, grid, iters), run<16, true>(din, dout, grid, iters),
           run<32, false>(din, dout, grid, iters), run<16, false>(din, dout, grid, iters));
  }
  printf("%s data: 32x32x16, 64 rows per wave (each LDS fragment feeds two MFMAs, 2 workgroups per CU) + softmax VALU %7.1f TF | bare %7.1f TF\n",
         zeros ? "zero  " : "random", run2<true>(din, dout, 512, iters / 2), run2<false>(din, dout, 512, iters / 2));
  return 0;
}
