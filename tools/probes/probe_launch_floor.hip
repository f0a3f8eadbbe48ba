// probe_launch_floor.hip -- what a back-to-back launch costs before any attention arithmetic (the floor under the short-sequence
// rows of the sweep): per-launch time, measured with HIP events around 200 launches on one stream, of
//   (a) an empty kernel, (b) a kernel that reads one 16-byte chunk per lane and writes it back (one dependent HBM/L2 round trip),
//   (c) two dependent round trips (load -> load at an address derived from the first -> store), (d) three;
// each with the grid of the N = 128 ... 1024 sweep rows (64 ... 512 workgroups of 256 threads, 32 KiB of dynamic LDS like the
// forward kernel). The forward's chain is: Q fragments + first K/V tile (one round trip, both in flight together), its arithmetic,
// the O / LSE stores (drained before the next launch may start).
// build: hipcc -O3 --offload-arch=gfx950 probe_launch_floor.hip -o probe_launch_floor ; run: ./probe_launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int HOPS>
__global__ __launch_bounds__(256) void chain(const uint4 *in, uint4 *out, unsigned mask) {
  extern __shared__ char smem[];
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (HOPS == 0) return;
  uint4 v = in[i & mask];
#pragma unroll
  for (int h = 1; h < HOPS; ++h) v = in[(v.x + i) & mask];
  if (threadIdx.x == 0) smem[0] = (char)v.y;
  out[i & mask] = v;
}

template <int HOPS>
static float time_one(int blocks, const uint4 *in, uint4 *out, unsigned mask) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(chain<HOPS>, dim3(blocks), dim3(256), 32768, 0, in, out, mask);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(chain<HOPS>, dim3(blocks), dim3(256), 32768, 0, in, out, mask);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best / 200.0f * 1000.0f;
}

int main() {
  const unsigned n = 1u << 20;  // 16 MiB of uint4
  uint4 *in, *out;
  (void)hipMalloc(&in, n * sizeof(uint4)); (void)hipMalloc(&out, n * sizeof(uint4));
  (void)hipMemset(in, 0, n * sizeof(uint4));
  printf("per-launch time, us (best of 5 x 200 back-to-back launches; 256 threads, 32 KiB dynamic LDS)\n");
  printf("%8s %10s %10s %10s %10s\n", "blocks", "empty", "1 trip", "2 trips", "3 trips");
  for (int blocks : {64, 128, 256, 512, 1024}) {
    printf("%8d %10.2f %10.2f %10.2f %10.2f\n", blocks, time_one<0>(blocks, in, out, n - 1), time_one<1>(blocks, in, out, n - 1),
           time_one<2>(blocks, in, out, n - 1), time_one<3>(blocks, in, out, n - 1));
  }
  return 0;
}
