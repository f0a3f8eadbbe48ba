// cpu_check.h -- the driver's own CPU correctness check and CPU timing baseline.
//
// The reference host verifies its kernels against loops written inline in main()
// (/root/reference/main.mm:121-159 non-causal, :549-578 causal). A drop-in driver keeps
// that check; this header is its restatement for the MI355X driver (fp32, same operation
// order per output element). It is part of the host program's verification step, never a
// compute fallback: the operator itself only exists on the GPU (include/fa_mi355.h).
// (The repository's test oracle under oracle/ is separate and is not linked here.)
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

namespace cpu {

// main.mm:24-30: every call re-seeds, so one seed gives one tensor; the reference uses 42
// for Q, K and V alike (Q == K == V). `seed` lets the driver draw independent tensors.
inline void init_random(float *data, size_t size, unsigned seed = 42) {
  std::mt19937 gen(seed);
  std::uniform_real_distribution<float> dis(-1.0f, 1.0f);
  for (size_t i = 0; i < size; ++i) data[i] = dis(gen);
}

// One query row. Non-causal follows main.mm:128-159 with the score row hoisted out of the
// per-d loop (same per-element order: max pass, then exp / num / den in j order); causal
// follows main.mm:551-577. `scores` is scratch of N floats.
inline void attention_row(const float *q, const float *k, const float *v, float *o, float *lse, int i, int N, int D,
                          float scale, bool causal, float *scores) {
  const int jn = causal ? i + 1 : N;
  const float *qi = q + (size_t)i * D;
  float max_s = -INFINITY;
  for (int j = 0; j < jn; ++j) {
    float score = 0.0f;
    const float *kj = k + (size_t)j * D;
    for (int d = 0; d < D; ++d) score += qi[d] * kj[d];
    score *= scale;
    scores[j] = score;
    if (score > max_s) max_s = score;
  }
  float den = 0.0f;
  for (int j = 0; j < jn; ++j) {
    scores[j] = (float)std::exp((double)(scores[j] - max_s));  // main.mm:153,567 call ::exp(double)
    den += scores[j];
  }
  for (int d = 0; d < D; ++d) {
    float num = 0.0f;
    for (int j = 0; j < jn; ++j) num += scores[j] * v[(size_t)j * D + d];
    o[(size_t)i * D + d] = num / den;
  }
  if (lse) lse[i] = max_s + (float)std::log((double)den);  // kernels.metal:862-864
}

// (Q,K,V,is_causal) -> O, LSE over contiguous [BH, N, D]; `threads` host threads over rows.
inline void attention(const float *q, const float *k, const float *v, float *o, float *lse, int BH, int N, int D,
                      float scale, bool causal, int threads) {
  threads = std::max(1, threads);
  std::atomic<long long> next{0};
  const long long rows = (long long)BH * N;
  auto work = [&]() {
    std::vector<float> scores((size_t)N);
    for (;;) {
      const long long r0 = next.fetch_add(8);
      if (r0 >= rows) break;
      for (long long r = r0; r < std::min(rows, r0 + 8); ++r) {
        const long long bh = r / N;
        const int i = causal ? (int)(N - 1 - r % N) : (int)(r % N);  // heavy causal rows first
        const size_t off = (size_t)bh * N * D;
        attention_row(q + off, k + off, v + off, o + off, lse ? lse + bh * N : nullptr, i, N, D, scale, causal,
                      scores.data());
      }
    }
  };
  if (threads == 1) {
    work();
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back(work);
  for (auto &t : pool) t.join();
}

// The reference's exact loop nest (main.mm:128-159): every output element recomputes all
// scores twice, O(N^2 D^2), single thread. Used only for the CPU timing table.
inline void attention_reference_structure(const float *q, const float *k, const float *v, float *o, int N, int D,
                                          float scale) {
  for (int i = 0; i < N; ++i)
    for (int d = 0; d < D; ++d) {
      float num = 0.0f, den = 0.0f, max_score = -INFINITY;
      for (int j = 0; j < N; ++j) {
        float score = 0.0f;
        for (int kk = 0; kk < D; ++kk) score += q[i * D + kk] * k[j * D + kk];
        score *= scale;
        if (score > max_score) max_score = score;
      }
      for (int j = 0; j < N; ++j) {
        float score = 0.0f;
        for (int kk = 0; kk < D; ++kk) score += q[i * D + kk] * k[j * D + kk];
        score *= scale;
        const float p = (float)std::exp((double)(score - max_score));
        num += p * v[j * D + d];
        den += p;
      }
      o[i * D + d] = num / den;
    }
}

// fp32 <-> 16-bit, round to nearest even (main.mm:322-329 casts through __fp16)
inline uint16_t f32_to_bf16(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
inline uint16_t f32_to_f16(float x) {
  _Float16 hv = (_Float16)x;
  uint16_t b;
  memcpy(&b, &hv, 2);
  return b;
}
inline float f16_to_f32(uint16_t b) {
  _Float16 hv;
  memcpy(&hv, &b, 2);
  return (float)hv;
}

}  // namespace cpu
