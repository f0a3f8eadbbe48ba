// Issue rates of the VALU instructions the softmax is made of, per SIMD, alone and mixed across waves.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/probe_valu_rates.hip -o tools/probes/probe_valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE per wave: 0 = v_exp_f32, 1 = v_fma_f32, 2 = v_add_f32, 3 = v_cvt_pk_bf16_f32, 4 = v_max3_f32, 5 = mfma 32x32x16 bf16
template <int MODE>
__device__ __forceinline__ void body(float (&x)[8], float c, int iters) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  if constexpr (MODE == 5) {
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)x[i]; b[i] = (__bf16)c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
    }
    x[0] = acc0[0] + acc1[3];
    return;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (MODE == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
        if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(c));
        if constexpr (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(c));
        if constexpr (MODE == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[k]) : "v"(c));
        if constexpr (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[k]) : "v"(c));
      }
    }
  }
}

// waves 0..3 of a workgroup land on SIMDs 0..3; waves 4..7 again on 0..3, ...: wave w runs MODE_A if (w / 4) is even, else MODE_B
template <int MODE_A, int MODE_B>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters, float c) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = out[threadIdx.x + i] * 1e-3f;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (((w >> 2) & 1) == 0) body<MODE_A>(x, c, iters); else body<MODE_B>(x, c, iters);
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

template <int A, int B>
static void run(const char *name, int waves, float *out, long long *cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 0.5f);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters, 0.5f);
  CHECK(hipDeviceSynchronize());
  long long h[16];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  const double nA = (A == 5) ? 8.0 * iters : 64.0 * iters, nB = (B == 5) ? 8.0 * iters : 64.0 * iters;
  printf("%-34s waves/SIMD %d : wave0 %.2f cyc/instr", name, waves / 4, (double)h[0] / nA);
  if (waves > 4) printf("   wave4 %.2f cyc/instr", (double)h[4] / nB);
  printf("\n");
}

int main() {
  float *out; long long *cyc;
  CHECK(hipMalloc(&out, 256 * 1024 * 4 + 64)); CHECK(hipMemset(out, 0, 256 * 1024 * 4 + 64));
  CHECK(hipMalloc(&cyc, 256 * 16 * 8));
  run<0, 0>("v_exp_f32", 4, out, cyc);       run<0, 0>("v_exp_f32", 8, out, cyc);
  run<1, 1>("v_fma_f32", 4, out, cyc);       run<1, 1>("v_fma_f32", 8, out, cyc);
  run<2, 2>("v_add_f32", 4, out, cyc);
  run<3, 3>("v_cvt_pk_bf16_f32", 4, out, cyc);
  run<4, 4>("v_max3_f32", 4, out, cyc);
  run<5, 5>("mfma_32x32x16_bf16", 4, out, cyc); run<5, 5>("mfma_32x32x16_bf16", 8, out, cyc);
  run<0, 1>("exp (w0) beside fma (w4)", 8, out, cyc);
  run<5, 0>("mfma (w0) beside exp (w4)", 8, out, cyc);
  run<5, 1>("mfma (w0) beside fma (w4)", 8, out, cyc);
  run<5, 3>("mfma (w0) beside cvt_pk (w4)", 8, out, cyc);
  return 0;
}
