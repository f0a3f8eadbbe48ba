// probe_tr8.hip -- hardware semantics the fp8 PV path relies on (exact data, printed as tables):
//  (1) ds_read_b64_tr_b8: which LDS bytes land in which byte of which lane, for lane l supplying address 8*l;
//  (2) v_cvt_pk_fp8_f32: byte placement (op_sel) and what happens above the e4m3 range.
// build: hipcc -O3 --offload-arch=gfx950 probe_tr8.hip -o probe_tr8 ; run: ./probe_tr8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

__global__ void probe(unsigned *out, unsigned *cv) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  lds_char *smem = (lds_char *)smem_g;
  const int lane = threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) smem[i] = (char)(pass == 0 ? (i & 0xff) : (i >> 8));
    __syncthreads();
    i32x2 r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2 *)(smem + 8 * lane));
    out[(pass * 64 + lane) * 2 + 0] = (unsigned)r[0];
    out[(pass * 64 + lane) * 2 + 1] = (unsigned)r[1];
  }
  if (lane == 0) {
    int w = 0x11223344;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(1.0f, 2.0f, w, false);
    cv[0] = (unsigned)w;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(0.5f, -3.0f, w, true);
    cv[1] = (unsigned)w;
    const float big[6] = {448.0f, 449.0f, 480.0f, 500.0f, 1.0e6f, __builtin_inff()};
    for (int i = 0; i < 6; ++i) cv[2 + i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(big[i], 0.0019f, 0, false);
    const float small[4] = {0.001953125f, 0.0009765625f, 0.00146484375f, 0.015625f};  // 2^-9, 2^-10, 1.5 * 2^-10, 2^-6
    for (int i = 0; i < 4; ++i) cv[8 + i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(small[i], 0.0f, 0, false);
  }
}

int main() {
  unsigned *out, *cv;
  hipMalloc(&out, 256 * 4); hipMalloc(&cv, 64 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, out, cv);
  unsigned h[256], c[64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(c, cv, sizeof(c), hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b8, lane l supplies address 8*l: lane -> source byte addresses of its 8 result bytes\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int b = 0; b < 8; ++b) {
      const unsigned lo = (h[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff, hi = (h[(64 + l) * 2 + b / 4] >> (8 * (b % 4))) & 0xff;
      printf(" %4u", hi * 256 + lo);
    }
    printf("\n");
  }
  printf("cvt_pk_fp8_f32(1.0, 2.0, old=0x11223344, false) = %08x ; then (0.5, -3.0, true) = %08x\n", c[0], c[1]);
  printf("above the range (448, 449, 480, 500, 1e6, inf | second value 0.0019): %04x %04x %04x %04x %04x %04x\n", c[2] & 0xffff, c[3] & 0xffff, c[4] & 0xffff, c[5] & 0xffff, c[6] & 0xffff, c[7] & 0xffff);
  printf("small (2^-9, 2^-10, 1.5*2^-10, 2^-6): %02x %02x %02x %02x\n", c[8] & 0xff, c[9] & 0xff, c[10] & 0xff, c[11] & 0xff);
  return 0;
}
