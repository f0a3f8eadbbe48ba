// fa_mfma_pipe_kernel.hip -- head_dim 64 operator, software-pipelined inside the wave.
//
// Same math, tile shapes, LDS images and lane maps as fa_mfma_kernel.hip (which documents
// them and keeps serving head_dim 128); what changes is WHEN things run. Measured on
// MI355X, the plain loop (QK^T -> softmax -> PV per tile, in that order, per wave) pays the
// MFMA time and the VALU time one after the other: co-resident waves phase-lock, so the
// matrix pipe idles during every softmax (45 % busy) although the VALU port is the scarcer
// resource at head_dim 64. Here each wave keeps TWO score tiles in flight so that every
// MFMA has independent VALU work issued right behind it:
//
//   phase A:  S(t+1) = K(t+1).Q^T   [8 MFMA]  ||  P(t) = exp2(c*S(t) - c*m), row sum, pack to 16-bit
//   phase B:  O^T   += V(t)^T.P(t)  [8 MFMA]  ||  mx = rowmax S(t+1)
//
// K therefore runs one tile ahead of V in the LDS double buffers: during iteration t,
// Kbuf[(t+1)&1] holds K(t+1), Vbuf[t&1] holds V(t), the freshly loaded K(t+2) / V(t+1)
// are written to the other halves, one barrier per tile as before. The rescale decision of
// tile t+1 (m, alpha) is taken at the top of iteration t+1, after PV(t) has issued, and
// covers O, l exactly once (cdna guide T13 hazard). sched_barrier pins the interleave.
#include "fa_mfma_common.h"

namespace fa {

template <typename Tag, bool CAUSAL>
__global__ __launch_bounds__(NTHREADS, 2) void fwd_mfma_pipe_d64_kernel(Params p) {
  using M = MT<Tag>;
  using vec8 = typename M::vec8;
  using elem = typename M::elem;
  constexpr int D = 64;
  constexpr int RB = D * 2, CPR = D / 8, KS = D / 16, DB = D / 32;
  constexpr int TILE = BN * RB;
  constexpr int NCH = BN * CPR / NTHREADS;

  extern __shared__ __attribute__((aligned(16))) char smem_generic[];
  lds_char *smem = (lds_char *)smem_generic;
  lds_char *Kbuf = smem;             // [2][BN][RB], rows swizzled
  lds_char *Vbuf = smem + 2 * TILE;  // [2][BN][RB], rows swizzled

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  const int nQ = (p.N + BM - 1) / BM;
  int bh, qb;
  map_block<CAUSAL>(blockIdx.x, p.B * p.H, nQ, bh, qb);
  const long long base = (long long)(bh / p.H) * p.batch_stride + (long long)(bh % p.H) * p.head_stride;
  const int q0 = qb * BM;
  const int qw0 = q0 + wave * WM;
  const int qrow = qw0 + r;

  const unsigned head_bytes = (unsigned)p.N * RB;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.q + base), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.k + base), 0, head_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)((const elem *)p.v + base), 0, head_bytes, 0x00020000);

  vec8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * RB + (2 * ks + h) * 16, 0, 0));

  // per-lane LDS offsets (derivation: fa_mfma_kernel.hip)
  const int kx = (r >> 1) & 7;
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = r * RB + (((2 * ks + h) ^ kx) << 4);
  const int g1 = (lane >> 4) & 1, vq = (lane >> 2) & 3, vp = lane & 3;
  const int vx = ((vq >> 1) & 1) << 2;
  int voff[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
    voff[db] = (4 * h + vq) * RB + ((((4 * db) ^ vx) + 2 * g1 + (vp >> 1)) << 4) + 8 * (vp & 1);
  int st_g[NCH], st_k[NCH], st_v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NTHREADS;
    const int row = c / CPR, ch = c % CPR;
    st_g[i] = row * RB + ch * 16;
    st_k[i] = row * RB + ((ch ^ ((row >> 1) & 7)) << 4);
    st_v[i] = row * RB + ((ch ^ (((row >> 1) & 1) << 2)) << 4);
  }

  const int kv_end = CAUSAL ? min(p.N, q0 + BM) : p.N;
  const int nT = (kv_end + BN - 1) / BN;
  // tiles this wave computes on (kernels.metal:682 with Br = 32): tile t is past the wave's
  // last query row once t*BN > qw0 + WM - 1
  const int nTw = CAUSAL ? min(nT, (qw0 + WM - 1) / BN + 1) : nT;

  u32x4 kst[NCH], vst[NCH];
  auto load_k = [&](int t) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)t * TILE + st_g[i], 0, 0);
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)t * TILE + st_g[i], 0, 0);
  };
  auto write_k = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) lds_write_b128(Kbuf + buf * TILE + st_k[i], kst[i]);
  };
  auto write_v = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) lds_write_b128(Vbuf + buf * TILE + st_v[i], vst[i]);
  };

  // masked iff key > qrow (kernels.metal:748) or key >= N; only called on tiles that need it
  auto apply_mask = [&](f32x16 (&s)[2], int kv0) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      int lim = p.N - 1 - kv0 - 32 * kb - 4 * h;
      if (CAUSAL) lim = min(lim, qrow - kv0 - 32 * kb - 4 * h);
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = ((i & 3) + 8 * (i >> 2) > lim) ? -INFINITY : s[kb][i];
    }
  };
  auto needs_mask = [&](int kv0) { return (CAUSAL && (kv0 + BN - 1 > qw0)) || (kv0 + BN > p.N); };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.0f;
  float m = -INFINITY, l = 0.0f;
  float mx;  // row max (both lane halves) of the score tile waiting in `sc`
  const float c2 = p.scale * 1.4426950408889634f;

  // ---- prologue: K(0), V(0) and K(1) into LDS, S(0) into registers
  load_k(0);
  load_v(0);
  write_k(0);
  write_v(0);
  if (nT > 1) {
    load_k(1);
    write_k(1);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));  // retire the Q loads before the loop
  __syncthreads();

  f32x16 sA[2], sB[2];
  {
    vec8 kf[2][KS];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = __builtin_bit_cast(vec8, lds_read_b128(Kbuf + kb * 32 * RB + koff[ks]));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sA[kb][i] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sA[kb] = M::mfma(kf[kb][ks], qf[ks], sA[kb]);
    }
    if (needs_mask(0)) apply_mask(sA, 0);
    float a = fmaxf(sA[0][0], sA[1][0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) a = fmaxf(fmaxf(a, sA[0][i]), sA[1][i]);
    float lo, hi;
    half_pair(a, lo, hi);
    mx = fmaxf(lo, hi);
  }

  // One iteration. BUF = t & 1 is a compile-time constant (all LDS addresses are base + immediate).
  // sc holds the raw scores of tile t on entry; sn receives those of tile t+1.
  auto tile = [&](auto bufc, const int t, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    constexpr int buf = decltype(bufc)::value;
    const bool have1 = t + 1 < nT, have2 = t + 2 < nT;
    if (have2) load_k(t + 2);
    if (have1) load_v(t + 1);

    if (t < nTw) {
      const lds_char *Kn = Kbuf + (buf ^ 1) * TILE;  // K(t+1)
      const lds_char *Vt = Vbuf + buf * TILE;        // V(t)
      // ---- rescale decision for tile t: everything still at the old max is scaled exactly once
      const float m_new = fmaxf(m, mx);
      if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
        const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c2);
        l *= alpha;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
        m = m_new;
      }
      const float mc = m * c2;
      const bool next_act = t + 1 < nTw;
      vec8 pf[2][2];
      float ls[2] = {0.0f, 0.0f};  // row-sum partials, accumulated right behind the exps

      // Softmax numerator in three stages over chunks of 4 score registers (chunk i = registers
      // 4(i&3).. of key block i>>2). The stages of one chunk depend on each other, so a step issues
      // stage F of chunk i+1, stage E of chunk i and stage S of chunk i-1: independent work only
      // (with chunks run back to back the 4-wide fma -> exp -> add -> cvt chains stalled the wave).
      auto stF = [&](int i) {  // x = c*s - c*m
        const int kb = i >> 2, b0 = 4 * (i & 3);
        float mc_i = mc;
        asm volatile("" : "+v"(mc_i));  // zero-instruction anchor: keeps the fmas in this step
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[kb][b0 + j] = __builtin_fmaf(sc[kb][b0 + j], c2, -mc_i);
      };
      auto stE = [&](int i) {  // p = 2^x
        const int kb = i >> 2, b0 = 4 * (i & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[kb][b0 + j] = __builtin_amdgcn_exp2f(sc[kb][b0 + j]);
      };
      auto stS = [&](int i) {  // row sum; pack to 16 bit when a PV fragment (8 registers) is complete
        const int kb = i >> 2, b0 = 4 * (i & 3);
        ls[i & 1] += (sc[kb][b0] + sc[kb][b0 + 1]) + (sc[kb][b0 + 2] + sc[kb][b0 + 3]);
        asm volatile("" : "+v"(ls[i & 1]));
        if (i & 1) {
          const int st = (i & 3) >> 1;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kb][st][j] = (elem)sc[kb][8 * st + j];
        }
      };
      // Fragment reads are issued LA MFMAs ahead of their use and die at it (registers: the
      // version that held all 24 fragments live needed 256 VGPRs and spilled).
      constexpr int LA = 2;
      vec8 kf[8];            // K fragment of QK step i = (kb = i>>2, ks = i&3)
      s16x4 vlo[8], vhi[8];  // V^T fragment of PV step j = (kb = j>>2, st = (j>>1)&1, db = j&1)
      auto kread = [&](int i) { kf[i] = __builtin_bit_cast(vec8, lds_read_b128(Kn + (i >> 2) * 32 * RB + koff[i & 3])); };
      auto vread = [&](int j) {
        const lds_char *vb = Vt + (32 * (j >> 2) + 16 * ((j >> 1) & 1)) * RB + voff[j & 1];
        vlo[j] = lds_read_tr16(vb);
        vhi[j] = lds_read_tr16(vb + 8 * RB);
      };

      // Phase A and phase B must stay ONE basic block (sched_barrier only binds the scheduler
      // inside a block; with a branch in between LLVM hoisted the fma pass above phase A and
      // sank the exp pass below it), so the rare mask is a compile-time variant of the body.
      auto body = [&](auto nextc, auto maskc) {
        constexpr bool NEXT = decltype(nextc)::value, MASK = decltype(maskc)::value;
        if constexpr (NEXT) {
          // ---- phase A: S(t+1) = K(t+1).Q^T  ||  P(t)
#pragma unroll
          for (int i = 0; i < LA; ++i) kread(i);
          __builtin_amdgcn_sched_barrier(0);
          f32x16 zero;
#pragma unroll
          for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
          stF(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            sn[i >> 2] = M::mfma(kf[i], qf[i & 3], (i & 3) == 0 ? zero : sn[i >> 2]);  // C = 0 is an inline constant
            if (i + LA < 8) kread(i + LA);
            else vread(i + LA - 8);  // the last steps start the V^T stream of phase B
            stE(i);
            if (i + 1 < 8) stF(i + 1);
            if (i >= 1) stS(i - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          stS(7);
          if constexpr (MASK) {
            apply_mask(sn, (t + 1) * BN);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < LA; ++j) vread(j);
#pragma unroll
          for (int i = 0; i < 8; ++i) stF(i);
#pragma unroll
          for (int i = 0; i < 8; ++i) stE(i);
#pragma unroll
          for (int i = 0; i < 8; ++i) stS(i);
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase B: O^T += V(t)^T.P(t)  ||  mx = rowmax S(t+1)
        float a0 = -INFINITY, a1 = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kb = j >> 2, st = (j >> 1) & 1, db = j & 1;
          const s16x8 v8 = __builtin_shufflevector(vlo[j], vhi[j], 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[db] = M::mfma(__builtin_bit_cast(vec8, v8), pf[kb][st], oacc[db]);
          if (j + LA < 8) vread(j + LA);
          if constexpr (NEXT) {  // 4 of the 32 row-max terms of the next tile ride behind each MFMA
            a0 = fmaxf(fmaxf(a0, sn[0][2 * j]), sn[0][2 * j + 1]);
            a1 = fmaxf(fmaxf(a1, sn[1][2 * j]), sn[1][2 * j + 1]);
            asm volatile("" : "+v"(a0), "+v"(a1));  // keep them behind this MFMA
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        l += ls[0] + ls[1];
        if constexpr (NEXT) {
          float lo, hi;
          half_pair(fmaxf(a0, a1), lo, hi);
          mx = fmaxf(lo, hi);
        }
      };
      if (next_act) {
        if (needs_mask((t + 1) * BN))
          body(std::true_type{}, std::true_type{});
        else
          body(std::true_type{}, std::false_type{});
      } else {
        body(std::false_type{}, std::false_type{});
      }
    }
    if (have2) write_k(buf);       // K(t+2) -> the half K(t) lived in (consumed in iteration t-1)
    if (have1) write_v(buf ^ 1);   // V(t+1) -> the half V(t-1) lived in
    __syncthreads();
  };

  for (int t = 0; t < nT; t += 2) {
    tile(std::integral_constant<int, 0>{}, t, sA, sB);
    if (t + 1 < nT) tile(std::integral_constant<int, 1>{}, t + 1, sB, sA);
  }

  // ---- epilogue (as fa_mfma_kernel.hip): normalise, LSE, O tile -> LDS -> 16-byte row stores
  {
    float lo, hi;
    half_pair(l, lo, hi);
    l = lo + hi;
  }
  const float inv_l = 1.0f / l;
  if (p.lse != nullptr && h == 0 && qrow < p.N) p.lse[(long long)bh * p.N + qrow] = m * p.scale + logf(l);

  lds_char *Ot = smem + wave * (WM * RB);
#pragma unroll
  for (int db = 0; db < DB; ++db) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      elem e0 = (elem)(oacc[db][4 * g4 + 0] * inv_l), e1 = (elem)(oacc[db][4 * g4 + 1] * inv_l);
      elem e2 = (elem)(oacc[db][4 * g4 + 2] * inv_l), e3 = (elem)(oacc[db][4 * g4 + 3] * inv_l);
      u32x2 w;
      w[0] = (unsigned)__builtin_bit_cast(unsigned short, e0) | ((unsigned)__builtin_bit_cast(unsigned short, e1) << 16);
      w[1] = (unsigned)__builtin_bit_cast(unsigned short, e2) | ((unsigned)__builtin_bit_cast(unsigned short, e3) << 16);
      const int col_b = (32 * db + 8 * g4 + 4 * h) * 2;
      const int ch = (col_b >> 4) ^ (r & (CPR - 1));
      lds_write_b64(Ot + r * RB + (ch << 4) + (col_b & 15), w);
    }
  }
  __syncthreads();
  elem *Og = (elem *)p.o + base;
#pragma unroll
  for (int it = 0; it < WM * CPR / 64; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / CPR, ch = idx % CPR;
    const u32x4 vv = lds_read_b128(Ot + row * RB + ((ch ^ (row & (CPR - 1))) << 4));
    if (qw0 + row < p.N) *reinterpret_cast<u32x4 *>(Og + (long long)(qw0 + row) * D + ch * 8) = vv;
  }
}

template <typename Tag>
static hipError_t launch_pipe_dt(const Params &p, hipStream_t s) {
  const int nQ = (p.N + BM - 1) / BM;
  const size_t smem = 4 * BN * 64 * 2;
  if (p.is_causal)
    hipLaunchKernelGGL((fwd_mfma_pipe_d64_kernel<Tag, true>), dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, p);
  else
    hipLaunchKernelGGL((fwd_mfma_pipe_d64_kernel<Tag, false>), dim3(nQ * p.B * p.H), dim3(NTHREADS), smem, s, p);
  return hipGetLastError();
}

hipError_t launch_mfma_pipe_d64(const Params &p, int dtype, hipStream_t s) {
  return dtype == FA_DTYPE_F16 ? launch_pipe_dt<F16>(p, s) : launch_pipe_dt<BF16>(p, s);
}

}  // namespace fa
