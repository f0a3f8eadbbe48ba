// probe_dq_atomic_floor.hip -- the float-atomic traffic of a 5-product (single-pass) attention backward, alone.
//
// In that form one workgroup owns Kw keys of a (batch, head) and walks the 32-row query slices that see them; for every slice it adds
// its [32 x D] fp32 share of dQ into global memory (cdna_hip_programming.md, "Attention backward"). This probe issues
// exactly those adds -- same addresses, same order, same causal trip counts, nothing else (no loads, no MFMAs) -- for the config-3 shape
// (B = 4, H = 16, N = 4096, D = 64, causal) and for its head_dim-128 twin, with Kw = 128 / 256 / 512, and prints the time per pass.
// That time is a FLOOR for a kernel of that form (its matrix work can at best hide behind it); the two-kernel, atomic-free backward of
// this library takes 421-433 us on the head_dim-64 shape and ~850 us on the head_dim-128 one.
// Access shape: one wave-instruction adds 64 consecutive floats (256 contiguous bytes: the full-rate shape of MI355X_MICROARCH.md,
// "Global float atomics"); no-return global_atomic_add_f32 (-munsafe-fp-atomics).
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics probe_dq_atomic_floor.hip -o probe_dq_atomic_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

// grid: (N / Kw) key blocks x BH; block: 256 threads = 4 waves; slice s of the block: queries 32 s .. 32 s + 31
__global__ __launch_bounds__(256) void dq_adds(float *dq, int N, int D, int Kw, int causal) {
  const int nkb = N / Kw, kb = blockIdx.x % nkb, bh = blockIdx.x / nkb;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *base = dq + (size_t)bh * N * D;
  const int s_begin = causal ? (kb * Kw) / 32 : 0, s_end = N / 32;
  const int per_slice = 32 * D / 64;  // wave-instructions of 64 floats per slice
  for (int s = s_begin; s < s_end; ++s) {
    float *tile = base + (size_t)s * 32 * D;
    for (int i = wave; i < per_slice; i += 4) atomicAdd(tile + i * 64 + lane, 1.0f);
  }
}

static float time_us(float *dq, int BH, int N, int D, int Kw, int causal) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = (N / Kw) * BH;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(dq_adds, dim3(grid), dim3(256), 0, 0, dq, N, D, Kw, causal);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(dq_adds, dim3(grid), dim3(256), 0, 0, dq, N, D, Kw, causal);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best / 5.0f * 1000.0f;
}

int main() {
  const int B = 4, H = 16, N = 4096;
  float *dq;
  (void)hipMalloc(&dq, (size_t)B * H * N * 128 * sizeof(float));
  (void)hipMemset(dq, 0, (size_t)B * H * N * 128 * sizeof(float));
  printf("dQ float-atomic traffic of a single-pass backward, alone (B=%d H=%d N=%d; 5 passes per timing, best of 5)\n", B, H, N);
  printf("%5s %7s %6s %12s %10s %10s   %s\n", "D", "causal", "Kw", "added MB", "us/pass", "TB/s", "two-kernel backward of this library");
  for (int D : {64, 128})
    for (int causal : {1, 0})
      for (int Kw : {128, 256, 512}) {
        double slices = 0;
        for (int kb = 0; kb < N / Kw; ++kb) slices += (N / 32) - (causal ? kb * Kw / 32 : 0);
        const double mb = slices * B * H * 32.0 * D * 4 / 1e6;
        const float us = time_us(dq, B * H, N, D, Kw, causal);
        printf("%5d %7d %6d %12.1f %10.1f %10.2f   %s\n", D, causal, Kw, mb, us, mb / us,
               D == 64 ? (causal ? "421-433 us" : "~780 us") : (causal ? "~850 us" : "~1530 us"));
      }
  return 0;
}
