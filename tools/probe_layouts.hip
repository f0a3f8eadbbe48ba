// probe_layouts.hip -- exact-integer probes of the gfx950 lane maps the MFMA kernel relies on.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o tools/probe_layouts ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_swap(unsigned *out) {
  unsigned u = threadIdx.x + 100;
  auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  out[threadIdx.x * 2] = sw[0];
  out[threadIdx.x * 2 + 1] = sw[1];
}
// C = A(32x16) * B(16x32), lane (r,h) element j: A[r][8h+j], B[8h+j][r]
__global__ void k_mfma(const float *A, const float *B, float *C) {
  int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; C[row * 32 + r] = c[i]; }
}
// tr read: LDS [16 rows][64 cols] of shorts, value = row*100+col; lane per the kernel's formula (unswizzled)
__global__ void k_tr(short *out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  int lane = threadIdx.x, h = lane >> 5, g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  int row = 4 * h + vq, col = 16 * g1 + 4 * vp;  // db = 0
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
      (__attribute__((address_space(3))) char *)lds + (row * 64 + col) * 2));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = t[e];
}
int main() {
  unsigned *dsw; hipMalloc(&dsw, 128 * 4);
  k_swap<<<1, 64>>>(dsw);
  std::vector<unsigned> sw(128); hipMemcpy(sw.data(), dsw, 512, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("swap lane %2d: in=%u sw0=%u sw1=%u\n", l, l + 100, sw[2 * l], sw[2 * l + 1]);
  std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 2 + j * 7) % 5 - 2);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
  float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
  printf("mfma 32x32x16 bf16 layout mismatches: %d of 1024\n", bad);
  short *dt; hipMalloc(&dt, 256 * 2);
  k_tr<<<1, 64>>>(dt);
  std::vector<short> t(256); hipMemcpy(t.data(), dt, 512, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 5, 15, 16, 17, 31, 32, 33, 48, 63})
    printf("tr lane %2d: %d %d %d %d   (expect col %d, rows %d..%d)\n", l, t[4 * l], t[4 * l + 1], t[4 * l + 2], t[4 * l + 3],
           l & 31, 4 * (l >> 5), 4 * (l >> 5) + 3);
  printf("hip err: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
