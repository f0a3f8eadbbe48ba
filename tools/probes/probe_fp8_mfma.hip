// Exact-integer probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit (E8M0 = 127) scales:
// which (lane, byte) of the 8-VGPR A / B operands holds which matrix element? Hypothesis under test:
//   A: lane l (r = l & 31, h = l >> 5), byte b (0..31) = A[row r][k = 32 h + b];  B likewise with column r.
// Prints PASS when D = A.B (CPU, integers) is reproduced exactly with asymmetric small-integer data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const unsigned char *A /*[32][64]*/, const unsigned char *B /*[64][32]*/, float *Dm /*[32][32]*/, int hyp) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  unsigned char a[32], b[32];
  for (int i = 0; i < 32; ++i) {
    const int kk = (hyp == 0) ? 32 * h + i : (hyp == 1) ? 16 * h + (i & 15) + 32 * (i >> 4) : 8 * h + (i & 7) + 16 * (i >> 3);
    a[i] = A[r * 64 + kk];
    b[i] = B[kk * 32 + r];
  }
  i32x8 av, bv;
  for (int w = 0; w < 8; ++w) {
    av[w] = a[4 * w] | (a[4 * w + 1] << 8) | (a[4 * w + 2] << 16) | (a[4 * w + 3] << 24);
    bv[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0 /*A: fp8*/, 0 /*B: fp8*/, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int n = 0; n < 16; ++n) Dm[((n & 3) + 8 * (n >> 2) + 4 * h) * 32 + r] = c[n];
}

static unsigned char enc(int v) {  // small integers in OCP e4m3
  static const unsigned char t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};  // 0, 1, 2, 3, 4
  return v < 0 ? (t[-v] | 0x80) : t[v];
}
int main() {
  unsigned char hA[32 * 64], hB[64 * 32];
  int iA[32 * 64], iB[64 * 32];
  for (int i = 0; i < 32; ++i) for (int kx = 0; kx < 64; ++kx) { iA[i * 64 + kx] = ((i * 3 + kx * 5) % 7) - 3; hA[i * 64 + kx] = enc(iA[i * 64 + kx]); }
  for (int kx = 0; kx < 64; ++kx) for (int j = 0; j < 32; ++j) { iB[kx * 32 + j] = ((kx * 2 + j * 3 + kx / 7) % 5) - 2; hB[kx * 32 + j] = enc(iB[kx * 32 + j]); }
  unsigned char *dA, *dB; float *dD; float hD[32 * 32];
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 3; ++hyp) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kx = 0; kx < 64; ++kx) s += iA[i * 64 + kx] * iB[kx * 32 + j]; if ((float)s != hD[i * 32 + j]) ++bad; }
    printf("hypothesis %d (%s): %s (%d of 1024 wrong)\n", hyp, hyp == 0 ? "k = 32h + b" : hyp == 1 ? "k = 16h + (b&15) + 32(b>>4)" : "k = 8h + (b&7) + 16(b>>3)", bad ? "FAIL" : "PASS", bad);
  }
  return 0;
}
