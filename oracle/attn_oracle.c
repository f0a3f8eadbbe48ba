/*
 * attn_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see attn_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * No -ffast-math, no FMA contraction: the golden vectors in tests/golden/ were
 * produced by the reference's own loops compiled with plain g++ -O2 and this
 * file must match them bit for bit.
 */
#include "attn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* mt19937 + libstdc++ uniform_real_distribution<float>(-1,1)                 */
/* follows /root/reference/main.mm:24-30 (std::mt19937 gen(42); dis(-1,1))    */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t mt[624];
  int idx;
} mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t x = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) x ^= 0x9908b0dfu;
      g->mt[i] = x;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

void oracle_init_random(float *data, long long size, uint32_t seed) {
  mt19937_t g;
  mt_seed(&g, seed);
  for (long long i = 0; i < size; ++i) {
    /* std::generate_canonical<float,24>: one 32-bit draw, float(u) / 2^32,
     * clamped below 1; then (b - a) * c + a with a=-1, b=1. */
    float sum = (float)mt_next(&g);
    float c = sum / 4294967296.0f;
    if (c >= 1.0f) c = nextafterf(1.0f, 0.0f);
    data[i] = c * (1.0f - (-1.0f)) + (-1.0f);
  }
}

/* ------------------------------------------------------------------------- */
/* main.mm:128-159 -- non-causal, every (i,d) recomputes the scores           */
/* ------------------------------------------------------------------------- */
void oracle_noncausal_faithful(const float *q, const float *k, const float *v,
                               float *o, int N, int D, float scale) {
  for (int i = 0; i < N; ++i) {
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
        /* main.mm:153 calls unqualified exp() on a float: the double overload
         * under g++/glibc. */
        float p = (float)exp((double)(score - max_score));
        num += p * v[j * D + d];
        den += p;
      }
      o[i * D + d] = num / den;
    }
  }
}

/* one query row, non-causal, hoisted: same per-element op order as above */
static void row_noncausal(const float *qi, const float *k, const float *v,
                          float *oi, float *lse_i, int N, int D, float scale,
                          float *scores) {
  float max_score = -INFINITY;
  for (int j = 0; j < N; ++j) {
    float score = 0.0f;
    const float *kj = k + (long long)j * D;
    for (int kk = 0; kk < D; ++kk) score += qi[kk] * kj[kk];
    score *= scale;
    scores[j] = score;
    if (score > max_score) max_score = score;
  }
  float den = 0.0f;
  for (int j = 0; j < N; ++j) {
    float p = (float)exp((double)(scores[j] - max_score));
    scores[j] = p;
    den += p;
  }
  for (int d = 0; d < D; ++d) {
    float num = 0.0f;
    for (int j = 0; j < N; ++j) num += scores[j] * v[(long long)j * D + d];
    oi[d] = num / den;
  }
  /* kernels.metal:862-864: L = m + log(l), fp32 */
  if (lse_i) *lse_i = max_score + (float)log((double)den);
}

/* one query row, causal: main.mm:551-577 */
static void row_causal(const float *qi, const float *k, const float *v,
                       float *oi, float *lse_i, int i, int D, float scale,
                       float *scores) {
  float max_s = -INFINITY;
  for (int j = 0; j <= i; ++j) {
    float score = 0.0f;
    const float *kj = k + (long long)j * D;
    for (int d = 0; d < D; ++d) score += qi[d] * kj[d];
    score *= scale;
    scores[j] = score;
    if (score > max_s) max_s = score;
  }
  float sum_exp = 0.0f;
  for (int j = 0; j <= i; ++j) {
    scores[j] = (float)exp((double)(scores[j] - max_s));
    sum_exp += scores[j];
  }
  for (int d = 0; d < D; ++d) {
    float val = 0.0f;
    for (int j = 0; j <= i; ++j) val += scores[j] * v[(long long)j * D + d];
    oi[d] = val / sum_exp;
  }
  if (lse_i) *lse_i = max_s + (float)log((double)sum_exp);
}

void oracle_noncausal_hoisted(const float *q, const float *k, const float *v,
                              float *o, float *lse, int N, int D, float scale) {
  float *scores = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
  for (int i = 0; i < N; ++i)
    row_noncausal(q + (long long)i * D, k, v, o + (long long)i * D,
                  lse ? lse + i : NULL, N, D, scale, scores);
  free(scores);
}

void oracle_causal(const float *q, const float *k, const float *v, float *o,
                   float *lse, int N, int D, float scale) {
  float *scores = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
  for (int i = 0; i < N; ++i)
    row_causal(q + (long long)i * D, k, v, o + (long long)i * D,
               lse ? lse + i : NULL, i, D, scale, scores);
  free(scores);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_attn_fwd(const float *q, const float *k, const float *v, float *o,
                     float *lse, int B, int H, int N, int D, float scale,
                     long long batch_stride, long long head_stride,
                     int is_causal, int threads) {
  long long rows = (long long)B * H * N;
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    float *scores = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long long r = 0; r < rows; ++r) {
      /* causal rows get heavier with i: walk them from the heavy end so the
       * dynamic schedule ends on light rows */
      long long bh = r / N;
      int i = is_causal ? (int)(N - 1 - (r % N)) : (int)(r % N);
      long long b = bh / H, h = bh % H;
      long long off = b * batch_stride + h * head_stride;
      /* LSE is [B,H,N] contiguous (kernels.metal:611,623) */
      float *lse_i = lse ? lse + bh * N + i : NULL;
      if (is_causal)
        row_causal(q + off + (long long)i * D, k + off, v + off,
                   o + off + (long long)i * D, lse_i, i, D, scale, scores);
      else
        row_noncausal(q + off + (long long)i * D, k + off, v + off,
                      o + off + (long long)i * D, lse_i, N, D, scale, scores);
    }
    free(scores);
  }
}

void oracle_attn_fwd_f64(const float *q, const float *k, const float *v,
                         double *o, double *lse, int B, int H, int N, int D,
                         float scale, long long batch_stride,
                         long long head_stride, int is_causal, int threads) {
  long long rows = (long long)B * H * N;
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    double *scores = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long long r = 0; r < rows; ++r) {
      long long bh = r / N;
      int i = is_causal ? (int)(N - 1 - (r % N)) : (int)(r % N);
      long long b = bh / H, h = bh % H;
      long long off = b * batch_stride + h * head_stride;
      const float *qi = q + off + (long long)i * D;
      int jmax = is_causal ? i : N - 1;
      double m = -INFINITY;
      for (int j = 0; j <= jmax; ++j) {
        const float *kj = k + off + (long long)j * D;
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += (double)qi[d] * (double)kj[d];
        s *= (double)scale;
        scores[j] = s;
        if (s > m) m = s;
      }
      double l = 0.0;
      for (int j = 0; j <= jmax; ++j) {
        scores[j] = exp(scores[j] - m);
        l += scores[j];
      }
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int j = 0; j <= jmax; ++j)
          acc += scores[j] * (double)v[off + (long long)j * D + d];
        o[off + (long long)i * D + d] = acc / l;
      }
      if (lse) lse[bh * N + i] = m + log(l);
    }
    free(scores);
  }
}

/* selected query rows of ONE head in fp64: lets the full-size parity tests
 * (N = 4096 .. 16384) check sampled rows without an O(N^2) pass per head */
void oracle_attn_rows_f64(const float *q, const float *k, const float *v,
                          double *o /*[nrows,D]*/, double *lse /*[nrows]*/,
                          int N, int D, float scale, int is_causal,
                          const int *rows, int nrows, int threads) {
  if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    double *scores = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int t = 0; t < nrows; ++t) {
      const int i = rows[t];
      const float *qi = q + (long long)i * D;
      int jmax = is_causal ? i : N - 1;
      double m = -INFINITY;
      for (int j = 0; j <= jmax; ++j) {
        const float *kj = k + (long long)j * D;
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += (double)qi[d] * (double)kj[d];
        s *= (double)scale;
        scores[j] = s;
        if (s > m) m = s;
      }
      double l = 0.0;
      for (int j = 0; j <= jmax; ++j) {
        scores[j] = exp(scores[j] - m);
        l += scores[j];
      }
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int j = 0; j <= jmax; ++j) acc += scores[j] * (double)v[(long long)j * D + d];
        o[(long long)t * D + d] = acc / l;
      }
      if (lse) lse[t] = m + log(l);
    }
    free(scores);
  }
}

/* ------------------------------------------------------------------------- */
/* generalised operator (scope row f3), fp64: grouped-query heads and Nq != Nk.  */
/* Query head h reads key/value head h / (Hq/Hkv). Causal with Nq != Nk is       */
/* bottom-right aligned: key j is visible to query i iff j <= i + (Nk - Nq)       */
/* (for Nq == Nk this is kernels.metal:748). Not in the reference: unpinned.      */
/* q,o: [B,Hq,Nq,D]; k,v: [B,Hkv,Nk,D]; lse: [B,Hq,Nq]; all contiguous.           */
/* ------------------------------------------------------------------------- */
void oracle_attn_fwd_ex_f64(const float *q, const float *k, const float *v, double *o, double *lse,
                            int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
                            int is_causal, int threads) {
  if (threads < 1) threads = 1;
  const int group = Hq / Hkv, off = Nk - Nq;
  const long long rows = (long long)B * Hq * Nq;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    double *scores = (double *)malloc(sizeof(double) * (size_t)(Nk > 0 ? Nk : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (long long r = 0; r < rows; ++r) {
      const long long bh = r / Nq;
      const int i = (int)(r % Nq);
      const long long b = bh / Hq, h = bh % Hq, hk = h / group;
      const float *qi = q + ((b * Hq + h) * Nq + i) * (long long)D;
      const float *K = k + (b * Hkv + hk) * (long long)Nk * D, *V = v + (b * Hkv + hk) * (long long)Nk * D;
      int jn = is_causal ? i + off + 1 : Nk;
      if (jn > Nk) jn = Nk;
      double m = -INFINITY, l = 0.0;
      for (int j = 0; j < jn; ++j) {
        double sc = 0.0;
        for (int d = 0; d < D; ++d) sc += (double)qi[d] * (double)K[(long long)j * D + d];
        sc *= (double)scale;
        scores[j] = sc;
        if (sc > m) m = sc;
      }
      for (int j = 0; j < jn; ++j) { scores[j] = exp(scores[j] - m); l += scores[j]; }
      double *oi = o + ((b * Hq + h) * Nq + i) * (long long)D;
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int j = 0; j < jn; ++j) acc += scores[j] * (double)V[(long long)j * D + d];
        oi[d] = jn > 0 ? acc / l : 0.0;
      }
      if (lse) lse[bh * Nq + i] = jn > 0 ? m + log(l) : -INFINITY;
    }
    free(scores);
  }
}

/* ------------------------------------------------------------------------- */
/* backward (fp64): dQ, dK, dV of the operator for upstream gradient dO.        */
/* Math of /root/reference/kernels.metal:905-1265 (D_i at :983-990, P = exp(S*scale - L_i)  */
/* at :1082-1089, dS = P*(dP - D_i)*scale at :1160-1169). The reference's own CPU check of   */
/* it is broken (main.mm:1100-1101 value-casts bit patterns) and never looks at dK/dV:       */
/* backward parity is UNPINNED by the reference; this fp64 statement is the anchor.          */
/* ------------------------------------------------------------------------- */
void oracle_attn_bwd_f64(const float *q, const float *k, const float *v, const float *d_o,
                         double *dq, double *dk, double *dv, int B, int H, int N, int D,
                         float scale, int is_causal, int threads) {
  if (threads < 1) threads = 1;
  const long long BH = (long long)B * H;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (long long bh = 0; bh < BH; ++bh) {
    const long long off = bh * (long long)N * D;
    const float *Q = q + off, *K = k + off, *V = v + off, *DO = d_o + off;
    double *DQ = dq + off, *DK = dk + off, *DV = dv + off;
    for (long long i = 0; i < (long long)N * D; ++i) DQ[i] = DK[i] = DV[i] = 0.0;
    double *p = (double *)malloc(sizeof(double) * (size_t)N);
    double *o = (double *)malloc(sizeof(double) * (size_t)D);
    for (int i = 0; i < N; ++i) {
      const int jn = is_causal ? i + 1 : N;
      double m = -INFINITY;
      for (int j = 0; j < jn; ++j) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += (double)Q[(long long)i * D + d] * (double)K[(long long)j * D + d];
        s *= (double)scale;
        p[j] = s;
        if (s > m) m = s;
      }
      double l = 0.0;
      for (int j = 0; j < jn; ++j) { p[j] = exp(p[j] - m); l += p[j]; }
      for (int j = 0; j < jn; ++j) p[j] /= l;
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int j = 0; j < jn; ++j) acc += p[j] * (double)V[(long long)j * D + d];
        o[d] = acc;
      }
      double delta = 0.0;  /* D_i = rowsum(dO * O) */
      for (int d = 0; d < D; ++d) delta += (double)DO[(long long)i * D + d] * o[d];
      for (int j = 0; j < jn; ++j) {
        double dp = 0.0;
        for (int d = 0; d < D; ++d) dp += (double)DO[(long long)i * D + d] * (double)V[(long long)j * D + d];
        const double ds = p[j] * (dp - delta) * (double)scale;
        for (int d = 0; d < D; ++d) {
          DV[(long long)j * D + d] += p[j] * (double)DO[(long long)i * D + d];
          DQ[(long long)i * D + d] += ds * (double)K[(long long)j * D + d];
          DK[(long long)j * D + d] += ds * (double)Q[(long long)i * D + d];
        }
      }
    }
    free(p);
    free(o);
  }
}

/* ------------------------------------------------------------------------- */
/* RNE casts (main.mm:322-329 does fp32 -> __fp16 with a C cast)              */
/* ------------------------------------------------------------------------- */
static uint32_t f32_bits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
static float bits_f32(uint32_t u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}

uint16_t oracle_f32_to_bf16_bits(float x) {
  uint32_t u = f32_bits(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float oracle_bf16_bits_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

uint16_t oracle_f32_to_f16_bits(float x) {
  uint32_t u = f32_bits(x);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         /* >= 65520 -> inf */
  if (a < 0x33000001u) return (uint16_t)sign;                      /* <= 2^-25 -> 0 */
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  int shift;                                /* bits to drop */
  uint32_t base;
  if (e < -14) { /* subnormal half: value = m * 2^(e-23), unit 2^-24 */
    shift = 13 + (-14 - e);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7fffffu;
  }
  uint32_t kept = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (kept & 1u))) kept += 1; /* carry may bump exponent */
  return (uint16_t)(sign | (base + kept));
}
float oracle_f16_bits_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  if (e == 0x1f) return bits_f32(sign | 0x7f800000u | (m << 13));
  if (e == 0) {
    float f = (float)m * 5.9604644775390625e-08f; /* 2^-24 */
    return sign ? -f : f;
  }
  return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

/* OCP e4m3fn: bias 7, max 448, no inf, NaN = S.1111.111; saturating. */
float oracle_fp8_e4m3_bits_to_f32(uint8_t b) {
  uint32_t e = (b >> 3) & 0xfu, m = b & 7u;
  float f;
  if (e == 0xf && m == 7) return NAN;
  if (e == 0) f = (float)m * 0.001953125f; /* 2^-9 */
  else f = ldexpf((float)(8u + m), (int)e - 10);
  return (b & 0x80u) ? -f : f;
}
uint8_t oracle_f32_to_fp8_e4m3_bits(float x) {
  uint8_t sign = (f32_bits(x) >> 24) & 0x80u;
  float a = fabsf(x);
  if (a != a) return (uint8_t)(sign | 0x7f);
  if (a >= 448.0f) return (uint8_t)(sign | 0x7e); /* saturate (464+ would round to 448 anyway) */
  if (a < 0.0009765625f) return sign;             /* < 2^-10 -> 0 (tie at 2^-10 -> even = 0) */
  int e;
  (void)frexpf(a, &e); /* a = f * 2^e, f in [0.5,1) -> exponent of leading bit = e-1 */
  int lead = e - 1;
  if (lead < -6) lead = -6;               /* subnormal: unit 2^-9 */
  float unit = ldexpf(1.0f, lead - 3);    /* 3 mantissa bits */
  float qv = a / unit;                    /* exact: power-of-two scaling */
  float r = nearbyintf(qv);               /* default mode = RNE */
  float val = r * unit;
  if (val >= 448.0f) return (uint8_t)(sign | 0x7e);
  /* encode val */
  if (val < 0.015625f) return (uint8_t)(sign | (uint8_t)(val / 0.001953125f));
  (void)frexpf(val, &e);
  int ee = e - 1 + 7;
  uint32_t mant = (uint32_t)(val / ldexpf(1.0f, e - 1 - 3)) - 8u;
  return (uint8_t)(sign | ((uint32_t)ee << 3) | mant);
}

void oracle_round_f16(float *x, long long n) {
  for (long long i = 0; i < n; ++i) x[i] = oracle_f16_bits_to_f32(oracle_f32_to_f16_bits(x[i]));
}
void oracle_round_bf16(float *x, long long n) {
  for (long long i = 0; i < n; ++i) x[i] = oracle_bf16_bits_to_f32(oracle_f32_to_bf16_bits(x[i]));
}
void oracle_round_fp8_e4m3(float *x, long long n) {
  for (long long i = 0; i < n; ++i)
    x[i] = oracle_fp8_e4m3_bits_to_f32(oracle_f32_to_fp8_e4m3_bits(x[i]));
}
