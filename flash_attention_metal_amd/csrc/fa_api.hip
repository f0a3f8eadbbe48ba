// fa_api.hip -- the C-ABI entry points of include/fa_mi355.h.
// Validation + dispatch only; every kernel lives in its own file.
#include <stdarg.h>
#include <algorithm>
#include <stdio.h>
#include <string.h>

#include "fa_common.h"

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace

extern "C" {

const char *fa_last_error(void) { return g_err; }
int fa_version(void) { return FA_MI355_VERSION; }

const char *fa_variant_name(int v) {
  switch (v) {
    case FA_VARIANT_AUTO: return "auto";
    case FA_VARIANT_NAIVE: return "naive";
    case FA_VARIANT_TILED: return "tiled";
    case FA_VARIANT_TILED_V2: return "tiled_v2";
    case FA_VARIANT_MFMA: return "mfma";
    case FA_VARIANT_MFMA_PP: return "mfma_pp";
    case FA_VARIANT_MFMA_SPLITKV: return "mfma_splitkv";
    case FA_VARIANT_MFMA_SPLIT2: return "mfma_split2";
    case FA_VARIANT_MFMA_EXACT: return "mfma_exact";
    case FA_VARIANT_MFMA_H64S2: return "mfma_h64s2";
    case FA_VARIANT_MFMA16: return "mfma16";
    case FA_VARIANT_MFMA_FP8PV: return "mfma_fp8pv";
    default: return "?";
  }
}
const char *fa_dtype_name(int d) {
  switch (d) {
    case FA_DTYPE_F32: return "f32";
    case FA_DTYPE_F16: return "f16";
    case FA_DTYPE_BF16: return "bf16";
    case FA_DTYPE_FP8_E4M3: return "fp8_e4m3";
    default: return "?";
  }
}
int fa_dtype_in_bytes(int d) {
  switch (d) {
    case FA_DTYPE_F32: return 4;
    case FA_DTYPE_F16: case FA_DTYPE_BF16: return 2;
    case FA_DTYPE_FP8_E4M3: return 1;
    default: return 0;
  }
}
int fa_dtype_out_bytes(int d) { return d == FA_DTYPE_FP8_E4M3 ? 2 : fa_dtype_in_bytes(d); }

int fa_supported(int dtype, int variant, int D) {
  switch (variant) {
    case FA_VARIANT_AUTO: return fa_resolve_variant(dtype, D) > 0;
    case FA_VARIANT_NAIVE: return fa::naive_supported(dtype, D);
    case FA_VARIANT_TILED: return fa::tiled_supported(dtype, D);
    case FA_VARIANT_TILED_V2: return fa::tiled_v2_supported(dtype, D);
    case FA_VARIANT_MFMA: return fa::mfma_supported(dtype, D);
    case FA_VARIANT_MFMA_PP: return 0;  // retired in round 4 (tools/experiments/fa_fwd_pp_kernel.hip): the enum value stays reserved
    case FA_VARIANT_MFMA_SPLITKV: return fa::splitkv_supported(dtype, D);
    case FA_VARIANT_MFMA_SPLIT2: return fa::mfma_split2_supported(dtype, D);
    case FA_VARIANT_MFMA_EXACT: return fa::mfma_supported(dtype, D);
    case FA_VARIANT_MFMA_H64S2: return fa::mfma_h64s2_supported(dtype, D);
    case FA_VARIANT_MFMA16: return fa::mfma16_supported(dtype, D);
    case FA_VARIANT_MFMA_FP8PV: return fa::fp8pv_supported(dtype, D);
    default: return 0;
  }
}
int fa_resolve_variant(int dtype, int D) {
  if (fa::mfma_supported(dtype, D)) return FA_VARIANT_MFMA;
  if (fa::mfma16_supported(dtype, D)) return FA_VARIANT_MFMA16;  // 16-bit inputs, the other multiples of 8 up to 128: zero-padded rows
  if (fa::tiled_v2_supported(dtype, D)) return FA_VARIANT_TILED_V2;
  return FA_ERR_UNSUPPORTED;
}

// AUTO between the matrix-core kernels (interleaved A/B on MI355X): the split-KV kernel wins on
// grids far smaller than the chip, the 64-row two-split form and the eight-wave form of the 128-row kernel on grids of up to
// two / one workgroup(s) per CU; everywhere else the 128-row kernel is fastest.
int fa_resolve_variant_for(int dtype, int D, int B, int H, int N, int is_causal) {
  const int v = fa_resolve_variant(dtype, D);
  if (v != FA_VARIANT_MFMA) return v;
  // (round 2 sent long head_dim-128 sequences to the paired-block kernel; since the 128-row kernel stages its tiles by
  // LDS-DMA it is 8-10 % ahead there too -- config 4 shard 1295 vs 1181 TFLOP/s, profiles/r03/ab_dma_late_and_d128_auto.log --
  // and round 4 retired FA_VARIANT_MFMA_PP: the kernel is kept under tools/experiments/)
  // small grids: fewer 128-row workgroups than a quarter of the CUs (or half, when each would walk >= 32 tiles):
  // split the keys of every 32-row block over the waves of a workgroup instead (config 2: 15.9 -> 10.6 us)
  const long long blocks128 = (long long)B * H * ((N + 127) / 128);
  // (head_dim 64 only: the head_dim-128 instantiation needs more than the 256 registers its eight-wave workgroup leaves a wave and
  // spills 820 B -- 8 heads x 1024: 73 us against 17-20 us for the kernels below, profiles/r03/ab_d128_small_grids.log; it stays
  // reachable by name)
  if (D == 64 && fa::splitkv_supported(dtype, D) && N > 64 && blocks128 <= 64) return FA_VARIANT_MFMA_SPLITKV;
  // head_dim 64, 16-bit inputs, at most two 64-row workgroups per CU: 64-row blocks whose wave pairs take the even / odd
  // tiles -- twice the workgroups and half the sequential tiles (h=8, N=2048 causal: 20.0 (eight-wave form) -> 17.9 us;
  // 64 heads x 256: 6.6 -> 5.9 us; profiles/r03/ab_h64s2.log)
  const long long blocks64 = (long long)B * H * ((N + 63) / 64);
  if (fa::mfma_h64s2_supported(dtype, D) && N >= 256 && blocks64 <= 512) return FA_VARIANT_MFMA_H64S2;
  // up to one 128-row workgroup per CU: the block's tiles are the critical path -- eight waves, even / odd tiles
  // (h=32, N=1024 causal: 16.1 -> 14.2 us; h=8, N=2048: 27.3 -> 22.0 us, split-KV 25.1)
  // (e4m3 inputs at head_dim 128: the all-fp8 kernel is ahead of the eight-wave form except on small grids of long sequences --
  // 32 heads x 512 / 2048: 11.4 -> 10.3 / 36.8 -> 32.7 us, 8 heads x 4096: 50.6 vs 53.6; profiles/r04/auto_check_late.log)
  const bool fp8_d128 = dtype == FA_DTYPE_FP8_E4M3 && D == 128;
  if (fa::mfma_split2_supported(dtype, D) && N >= 512 && blocks128 <= 256 && !(fp8_d128 && N < 4096)) return FA_VARIANT_MFMA_SPLIT2;
  // causal, two workgroups' worth of blocks per CU of which half are light: the heaviest block's 32+ tiles are still the critical
  // path (32 heads x 2048: 28.9 -> 25.9 us at head_dim 64, 45.8 -> 42.5 at 128, fp8 29.3 -> 24.6; at N = 1024 the plain kernel
  // wins again; profiles/r03/auto_check.log)
  // (fp8 inputs at head_dim 64 already from N = 1024: 64 heads x 1024, 16.7 -> 14.7 us)
  const int n_min = (dtype == FA_DTYPE_FP8_E4M3 && D == 64) ? 1024 : 2048;
  if (fa::mfma_split2_supported(dtype, D) && is_causal && N >= n_min && blocks128 <= 512 && !fp8_d128) return FA_VARIANT_MFMA_SPLIT2;
  // head_dim 64, 16-bit inputs, grids that fill the chip: the same workgroup on the 16x16x32 instruction with its row sums on the matrix
  // core -- config 3 +2 %, N >= 4096 non-causal / N >= 8192 causal +4.5..5.7 %; level at N = 2048, 1-2 % behind at N = 1024 (its
  // first tile pays a second score pass) (profiles/r04/ab_mfma16_ones_vs_adds.log)
  // head_dim 128: from N = 8192 on (config 4 shard 1259-1300 -> 1323-1331 TFLOP/s, non-causal N = 8192 +4..5.7 %; level at N <= 4096;
  // profiles/r04/ab_mfma16_d128*.log)
  // (late round 4, on warm clocks, profiles/r04/ab_auto_short_sequences_mfma16.log: with its first tile free of the second score pass it is
  // also ahead from N = 1536 on under the mask (+4 %; level at 1024) and, without the mask, from N = 512 on for grids of at least 512
  // workgroups (+2..6 %, eight waves per workgroup there: mfma16_waves))
  if (fa::mfma16_supported(dtype, D)) {
    // (head_dim 128: since the first tile lost its second score pass it is ahead from N = 2048 on, +5.5..7 % at 4 x 16 heads x 2048 / 4096
    // causal and non-causal; without the mask from 1024 on, +4..5 %; auto_check_late.log)
    const bool take = D == 64 ? (is_causal ? N >= 1536 : (N >= 2048 || (N >= 512 && blocks128 >= 512))) : (is_causal ? N >= 2048 : N >= 1024);
    if (take) return FA_VARIANT_MFMA16;
  }
  // fp8 inputs, head_dim 64, grids that fill the chip: both products on the fp8 matrix pipe (config 5: 1215-1245 -> 1364-1461 TFLOP/s,
  // profiles/r04/ab_fp8pv_*.log). Its probabilities are e4m3 (include/fa_mi355.h, "fp8 probabilities"); FA_VARIANT_MFMA keeps them bf16
  if (fa::fp8pv_supported(dtype, D)) return FA_VARIANT_MFMA_FP8PV;
  return FA_VARIANT_MFMA;
}

const char *fa_fwd_kernel_name(int dtype, int D, int B, int H, int N, int is_causal) {
  static thread_local char name[96];
  const int v = fa_resolve_variant_for(dtype, D, B, H, N, is_causal);
  const char *tag = dtype == FA_DTYPE_F32 ? "float" : dtype == FA_DTYPE_F16 ? "fa::F16" : dtype == FA_DTYPE_BF16 ? "fa::BF16" : "fa::FP8";
  const char *c = is_causal ? "true" : "false";
  switch (v) {
    case FA_VARIANT_MFMA_SPLITKV: snprintf(name, sizeof(name), "fa::fwd_splitkv_kernel<%s, %d, %s>", tag, D, c); break;
    case FA_VARIANT_MFMA_SPLIT2: snprintf(name, sizeof(name), "fa::fwd_mfma_split2_kernel<%s, %d, %s>", tag, D, c); break;
    case FA_VARIANT_MFMA_H64S2: snprintf(name, sizeof(name), "fa::fwd_mfma_h64s2_kernel<%s, %d, %s>", tag, D, c); break;
    case FA_VARIANT_MFMA16:
      // (template arguments as rocprofv3 prints them: dtype, head dim of the instantiation, causal, waves per workgroup, padded rows, tiles per barrier)
      snprintf(name, sizeof(name), "fa::fwd_mfma16_kernel<%s, %d, %s, %d, %s, 1>", tag, (D == 64 || D == 128) ? D : (D < 64 ? 64 : 128), c,
               fa::mfma16_waves(D, B * H, N, N, is_causal), (D == 64 || D == 128) ? "false" : "true");
      break;
    case FA_VARIANT_MFMA_FP8PV: snprintf(name, sizeof(name), "fa::fwd_fp8_kernel<%d, %s>", D, c); break;
    case FA_VARIANT_MFMA:
      snprintf(name, sizeof(name), "fa::fwd_mfma_kernel<%s, %d, %s, %s>", tag, D, c,
               (dtype != FA_DTYPE_FP8_E4M3 && D <= 128) ? "true" : "false");  // pre-scaled operand where it exists
      break;
    case FA_VARIANT_MFMA_EXACT: snprintf(name, sizeof(name), "fa::fwd_mfma_kernel<%s, %d, %s, false>", tag, D, c); break;
    case FA_VARIANT_TILED_V2: snprintf(name, sizeof(name), "fa::tiled_v2_kernel"); break;
    default: name[0] = 0;
  }
  return name;
}

double fa_algorithmic_flops(int B, int H, int N, int D, int is_causal) {
  return (is_causal ? 2.0 : 4.0) * (double)B * H * (double)N * (double)N * D;
}
double fa_algorithmic_bytes(int B, int H, int N, int D, int dtype) {
  return (3.0 * fa_dtype_in_bytes(dtype) + fa_dtype_out_bytes(dtype)) * (double)B * H * N * D +
         4.0 * (double)B * H * N;
}

int fa_fwd(const void *q, const void *k, const void *v, void *o, float *lse, int B, int H, int N,
           int D, float scale, long long batch_stride, long long head_stride, int is_causal,
           int dtype, int variant, void *hip_stream) {
  g_err[0] = 0;
  if (!q || !k || !v || !o) return fail(FA_ERR_INVALID_ARG, "fa_fwd: null tensor pointer");
  if (B < 1 || H < 1 || N < 1 || D < 1)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: B=%d H=%d N=%d D=%d must be >= 1", B, H, N, D);
  if (!(scale > 0.0f)) return fail(FA_ERR_INVALID_ARG, "fa_fwd: scale=%g must be > 0", (double)scale);
  if (fa_dtype_in_bytes(dtype) == 0) return fail(FA_ERR_INVALID_ARG, "fa_fwd: bad dtype %d", dtype);
  if (head_stride < (long long)N * D || batch_stride < 0 || (H > 1 && batch_stride < head_stride))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: strides (batch %lld, head %lld) smaller than a head (N*D=%lld)",
                batch_stride, head_stride, (long long)N * D);
  const int stride_mult = dtype == FA_DTYPE_FP8_E4M3 ? 16 : 8;  // keeps every head 16-byte aligned
  if ((batch_stride % stride_mult) || (head_stride % stride_mult))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: strides must be multiples of %d elements", stride_mult);
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: tensors must be 16-byte aligned");
  // 32-bit byte offsets inside a head; the staging loops may address up to two 64-key tiles past its end (range-checked
  // by the buffer descriptor, but the offset itself must not wrap)
  if ((double)(N + 128) * D * fa_dtype_in_bytes(dtype) >= 4294967296.0)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: one head exceeds 4 GiB");
  // head dims without a kernel of their own run on zero-padded rows whose padding is fetched from offset 2^31 + ... (fa_mfma16_kernel.hip, PAD)
  if (!fa::mfma_supported(dtype, D) && fa::mfma16_supported(dtype, D) && (double)(N + 128) * D * 2 >= 2147483648.0)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: one head exceeds 2 GiB (head dims on padded rows)");
  if ((long long)B * H > 0x7fffffffLL / ((N + 127) / 128))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: grid too large");
  if (variant == FA_VARIANT_AUTO) {
    variant = fa_resolve_variant_for(dtype, D, B, H, N, is_causal);
    if (variant < 0)
      return fail(FA_ERR_UNSUPPORTED, "fa_fwd: no kernel for dtype=%s D=%d", fa_dtype_name(dtype), D);
  }
  if (!fa_supported(dtype, variant, D))
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd: variant=%s does not support dtype=%s D=%d",
                fa_variant_name(variant), fa_dtype_name(dtype), D);
  if ((variant == FA_VARIANT_NAIVE || variant == FA_VARIANT_TILED || variant == FA_VARIANT_TILED_V2) &&
      (H > 65535 || B > 65535))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd: B,H must be <= 65535 for variant %s", fa_variant_name(variant));

  fa::Params p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
  p.B = B; p.H = H; p.N = N; p.D = D;
  p.scale = scale;
  p.batch_stride = batch_stride; p.head_stride = head_stride;
  p.is_causal = is_causal ? 1 : 0;
  p.Nk = N; p.Hkv = H;
  p.kv_batch_stride = batch_stride; p.kv_head_stride = head_stride;
  p.exact = (variant == FA_VARIANT_MFMA_EXACT);
  hipStream_t s = (hipStream_t)hip_stream;
  hipError_t e;
  switch (variant) {
    case FA_VARIANT_NAIVE: e = fa::launch_naive(p, dtype, s); break;
    case FA_VARIANT_TILED: e = fa::launch_tiled(p, dtype, s); break;
    case FA_VARIANT_TILED_V2: e = fa::launch_tiled_v2(p, dtype, s); break;
    case FA_VARIANT_MFMA_SPLITKV: e = fa::launch_splitkv(p, dtype, s); break;
    case FA_VARIANT_MFMA_SPLIT2: e = fa::launch_mfma_split2(p, dtype, s); break;
    case FA_VARIANT_MFMA_H64S2: e = fa::launch_mfma_h64s2(p, dtype, s); break;
    case FA_VARIANT_MFMA16: e = fa::launch_mfma16(p, dtype, s); break;
    case FA_VARIANT_MFMA_FP8PV: e = fa::launch_fp8pv(p, dtype, s); break;
    default: e = fa::launch_mfma(p, dtype, s); break;
  }
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
    return fail(FA_ERR_NO_DEVICE, "fa_fwd: %s", hipGetErrorString(e));
  if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_fwd: launch failed: %s", hipGetErrorString(e));
  return FA_OK;
}

int fa_fwd_ex(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk,
              int D, float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride,
              long long kv_head_stride, int is_causal, int dtype, void *hip_stream) {
  return fa_fwd_exv(q, k, v, o, lse, B, Hq, Hkv, Nq, Nk, D, scale, q_batch_stride, q_head_stride, kv_batch_stride, kv_head_stride,
                    is_causal, dtype, FA_VARIANT_AUTO, hip_stream);
}

int fa_fwd_exv(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk,
               int D, float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride,
               long long kv_head_stride, int is_causal, int dtype, int variant, void *hip_stream) {
  g_err[0] = 0;
  if (!q || !k || !v || !o) return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: null tensor pointer");
  if (B < 1 || Hq < 1 || Hkv < 1 || Nq < 1 || Nk < 1 || D < 1)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: sizes must be >= 1");
  if (Hq % Hkv) return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
  if (is_causal && Nk < Nq)
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_ex: causal needs Nk >= Nq (bottom-right alignment would leave empty rows)");
  if (!(scale > 0.0f)) return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: scale=%g must be > 0", (double)scale);
  const bool padded_dim = !fa::mfma_supported(dtype, D) && fa::mfma16_supported(dtype, D);  // (zero-padded rows of the 16x16x32 kernel)
  if (!fa::mfma_supported(dtype, D) && !padded_dim)
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_ex: needs a matrix-core kernel (f16/bf16: D a multiple of 8 up to 128, or 256; fp8: D=64|128|256), got dtype=%s D=%d",
                fa_dtype_name(dtype), D);
  if (padded_dim && (double)(std::max(Nq, Nk) + 128) * D * 2 >= 2147483648.0)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: one head exceeds 2 GiB (head dims on padded rows)");
  const int sm = dtype == FA_DTYPE_FP8_E4M3 ? 16 : 8;
  if (q_head_stride < (long long)Nq * D || kv_head_stride < (long long)Nk * D || (q_batch_stride % sm) || (q_head_stride % sm) ||
      (kv_batch_stride % sm) || (kv_head_stride % sm) || (Hq > 1 && B > 1 && q_batch_stride < q_head_stride) ||
      (Hkv > 1 && B > 1 && kv_batch_stride < kv_head_stride))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: bad strides");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: tensors must be 16-byte aligned");
  // (as in fa_fwd: the staging loops may address up to two 64-key tiles past the end of a head)
  if ((double)(std::max(Nq, Nk) + 128) * D * fa_dtype_in_bytes(dtype) >= 4294967296.0)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: one head exceeds 4 GiB");
  if (q_batch_stride < 0 || kv_batch_stride < 0) return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: negative batch stride");
  if ((long long)B * Hq > 0x7fffffffLL / ((Nq + 127) / 128)) return fail(FA_ERR_INVALID_ARG, "fa_fwd_ex: grid too large");
  fa::Params p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse;
  p.B = B; p.H = Hq; p.N = Nq; p.D = D; p.scale = scale;
  p.batch_stride = q_batch_stride; p.head_stride = q_head_stride;
  p.is_causal = is_causal ? 1 : 0;
  p.Nk = Nk; p.Hkv = Hkv; p.kv_batch_stride = kv_batch_stride; p.kv_head_stride = kv_head_stride;
  // few query blocks against many keys (decode steps, short prompts): the AUTO rule of fa_fwd for small grids -- the keys of every
  // 32-row block are split over the waves of a workgroup (the split-KV kernel takes the same Params: Nk, key/value heads).
  // head_dim 64 only: there it is 1.2-3x faster than the 128-row kernel at every key length (32 heads x 1 query x 16384 keys:
  // 165 -> 55 us), at head_dim 128 its four private 32-KiB tiles make it 2-4x SLOWER (profiles/r03/decode_steps_splitkv_rule.log)
  const long long blocks128 = (long long)B * Hq * ((Nq + 127) / 128);
  if (variant == FA_VARIANT_AUTO) {
    const bool small_grid = D == 64 && fa::splitkv_supported(dtype, D) && Nk > 64 && blocks128 <= 64;
    // the 16x16x32 kernel where fa_fwd's AUTO takes it: head_dim 64, 16-bit inputs, long key sequences on a grid that fills the chip
    variant = small_grid ? FA_VARIANT_MFMA_SPLITKV
              : padded_dim ? FA_VARIANT_MFMA16
              : (fa::mfma16_supported(dtype, D) && blocks128 > 512 &&
                 Nk >= (D == 64 ? (is_causal ? 1536 : 1024) : (is_causal ? 2048 : 1024)))  // (fa_fwd's thresholds, fa_resolve_variant_for)
                    ? FA_VARIANT_MFMA16
                    : FA_VARIANT_MFMA;
  }
  // the kernels that take the generalised problem (key/value heads, Nk): the 128-row kernel with / without its pre-scaled operand, its
  // 16x16x32 form, the split-KV kernel
  if (variant != FA_VARIANT_MFMA && variant != FA_VARIANT_MFMA_EXACT && variant != FA_VARIANT_MFMA16 && variant != FA_VARIANT_MFMA_SPLITKV)
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_ex: variant=%s does not take grouped heads / Nq != Nk (mfma, mfma_exact, mfma16, mfma_splitkv do)",
                fa_variant_name(variant));
  if (!fa_supported(dtype, variant, D))
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_ex: variant=%s does not support dtype=%s D=%d", fa_variant_name(variant), fa_dtype_name(dtype), D);
  p.exact = (variant == FA_VARIANT_MFMA_EXACT);
  hipStream_t s = (hipStream_t)hip_stream;
  const hipError_t e = variant == FA_VARIANT_MFMA_SPLITKV ? fa::launch_splitkv(p, dtype, s)
                       : variant == FA_VARIANT_MFMA16     ? fa::launch_mfma16(p, dtype, s)
                                                          : fa::launch_mfma(p, dtype, s);
  if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_fwd_ex: launch failed: %s", hipGetErrorString(e));
  return FA_OK;
}

int fa_fwd_decode_supported(int dtype, int D, int Hq, int Hkv, int Nq) {
  return fa::decode_supported(dtype, D) && Hq >= 1 && Hkv >= 1 && Nq >= 1 && Hq % Hkv == 0 && (long long)(Hq / Hkv) * Nq <= 32;
}
long long fa_fwd_decode_workspace_bytes(int B, int Hq, int Hkv, int Nq, int Nk, int D) {
  if (B < 1 || Hq < 1 || Hkv < 1 || Nq < 1 || Nk < 1 || (D != 64 && D != 128) || Hq % Hkv) return 0;
  return fa::decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D);
}
static int decode_impl(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk, int D,
                       float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
                       int is_causal, int dtype, int kv8, void *workspace, long long workspace_bytes, void *hip_stream);
int fa_fwd_decode(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk, int D,
                  float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
                  int is_causal, int dtype, void *workspace, long long workspace_bytes, void *hip_stream) {
  return decode_impl(q, k, v, o, lse, B, Hq, Hkv, Nq, Nk, D, scale, q_batch_stride, q_head_stride, kv_batch_stride, kv_head_stride, is_causal,
                     dtype, dtype == FA_DTYPE_FP8_E4M3, workspace, workspace_bytes, hip_stream);
}
int fa_fwd_decode_kv8(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk, int D,
                      float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
                      int is_causal, int q_dtype, void *workspace, long long workspace_bytes, void *hip_stream) {
  g_err[0] = 0;
  if (q_dtype != FA_DTYPE_BF16 && q_dtype != FA_DTYPE_FP8_E4M3)
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_decode_kv8: queries must be bf16 (or e4m3: then this is fa_fwd_decode), got %s", fa_dtype_name(q_dtype));
  return decode_impl(q, k, v, o, lse, B, Hq, Hkv, Nq, Nk, D, scale, q_batch_stride, q_head_stride, kv_batch_stride, kv_head_stride, is_causal,
                     q_dtype, 1, workspace, workspace_bytes, hip_stream);
}
static int decode_impl(const void *q, const void *k, const void *v, void *o, float *lse, int B, int Hq, int Hkv, int Nq, int Nk, int D,
                       float scale, long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
                       int is_causal, int dtype, int kv8, void *workspace, long long workspace_bytes, void *hip_stream) {
  g_err[0] = 0;
  if (!q || !k || !v || !o || !workspace) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: null pointer");
  if (B < 1 || Hq < 1 || Hkv < 1 || Nq < 1 || Nk < 1 || D < 1) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: sizes must be >= 1");
  if (Hq % Hkv) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: Hq=%d must be a multiple of Hkv=%d", Hq, Hkv);
  if (is_causal && Nk < Nq)
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_decode: causal needs Nk >= Nq (bottom-right alignment would leave empty rows)");
  if (!(scale > 0.0f)) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: scale=%g must be > 0", (double)scale);
  if (!fa_fwd_decode_supported(dtype, D, Hq, Hkv, Nq))
    return fail(FA_ERR_UNSUPPORTED, "fa_fwd_decode: needs f16 / bf16 / fp8_e4m3, D = 64 | 128 and (Hq / Hkv) * Nq <= 32 packed query rows; got dtype=%s D=%d "
                "Hq=%d Hkv=%d Nq=%d (use fa_fwd_ex)", fa_dtype_name(dtype), D, Hq, Hkv, Nq);
  const int dsm = dtype == FA_DTYPE_FP8_E4M3 ? 16 : 8, ksm = kv8 ? 16 : dsm;  // keeps every head 16-byte aligned
  if (q_head_stride < (long long)Nq * D || kv_head_stride < (long long)Nk * D || (q_batch_stride % dsm) || (q_head_stride % dsm) ||
      (kv_batch_stride % ksm) || (kv_head_stride % ksm) || q_batch_stride < 0 || kv_batch_stride < 0 ||
      (Hq > 1 && B > 1 && q_batch_stride < q_head_stride) || (Hkv > 1 && B > 1 && kv_batch_stride < kv_head_stride))
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: bad strides");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)workspace) & 15)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: tensors and workspace must be 16-byte aligned");
  if ((double)(Nk + 128) * D * 2 >= 4294967296.0) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: one head exceeds 4 GiB");
  const long long need = fa::decode_workspace_bytes(B, Hq, Hkv, Nq, Nk, D);
  if (workspace_bytes < need)
    return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: workspace of %lld bytes, fa_fwd_decode_workspace_bytes() asks for %lld", workspace_bytes, need);
  if ((long long)B * Hq * Nq > 0x7fffffffLL || (long long)B * Hkv * 256 > 0x7fffffffLL) return fail(FA_ERR_INVALID_ARG, "fa_fwd_decode: grid too large");
  fa::DecodeParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.ws = (float *)workspace;
  p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
  p.q_bs = q_batch_stride; p.q_hs = q_head_stride; p.kv_bs = kv_batch_stride; p.kv_hs = kv_head_stride;
  p.is_causal = is_causal ? 1 : 0;
  p.S = fa::decode_splits(B, Hkv, Nk, D, kv8);
  const hipError_t e = fa::launch_decode(p, D, dtype, kv8, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "fa_fwd_decode: launch failed: %s", hipGetErrorString(e));
  return FA_OK;
}

long long fa_bwd_workspace_bytes(int B, int H, int N) { return (long long)B * H * N * 4; }
// elements from the first to one past the last of a [B,H,N,D] tensor under (batch, head) strides, rounded up to 16
static long long extent16(int B, int H, int N, int D, long long bs, long long hs) {
  const long long e = (long long)(B - 1) * bs + (long long)(H - 1) * hs + (long long)N * D;
  return (e + 15) / 16 * 16;
}
static long long align256(long long x) { return (x + 255) / 256 * 256; }
long long fa_bwd_workspace_bytes_ex(int dtype, int B, int Hq, int Hkv, int Nq, int Nk, int D, long long q_batch_stride,
                                    long long q_head_stride, long long kv_batch_stride, long long kv_head_stride) {
  long long bytes = align256((long long)B * Hq * Nq * 4);
  if (dtype == FA_DTYPE_FP8_E4M3)  // bf16 copies of Q, K, V under their own strides
    bytes += align256(2 * extent16(B, Hq, Nq, D, q_batch_stride, q_head_stride)) +
             2 * align256(2 * extent16(B, Hkv, Nk, D, kv_batch_stride, kv_head_stride));
  return bytes;
}
int fa_bwd_supported(int dtype, int D) { return fa::bwd_supported(dtype, D); }
double fa_bwd_algorithmic_flops(int B, int H, int N, int D, int is_causal) {
  return 2.5 * fa_algorithmic_flops(B, H, N, D, is_causal);
}

static int bwd_impl(const char *fn, const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse,
                    float *dq, float *dk, float *dv, void *workspace, int B, int H, int Hkv, int N, int Nk, int D, float scale,
                    long long bs, long long hs, long long kbs, long long khs, int is_causal, int dtype, void *hip_stream) {
  g_err[0] = 0;
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !workspace)
    return fail(FA_ERR_INVALID_ARG, "%s: null pointer", fn);
  if (B < 1 || H < 1 || N < 1 || Nk < 1 || D < 1)
    return fail(FA_ERR_INVALID_ARG, "%s: B=%d H=%d Nq=%d Nk=%d D=%d must be >= 1", fn, B, H, N, Nk, D);
  if (is_causal && Nk < N)
    return fail(FA_ERR_UNSUPPORTED, "%s: causal needs Nk >= Nq (bottom-right alignment would leave empty rows)", fn);
  if (Hkv < 1 || H % Hkv) return fail(FA_ERR_INVALID_ARG, "%s: Hkv=%d must divide Hq=%d", fn, Hkv, H);
  if (!(scale > 0.0f)) return fail(FA_ERR_INVALID_ARG, "%s: scale=%g must be > 0", fn, (double)scale);
  const int smul = dtype == FA_DTYPE_FP8_E4M3 ? 16 : 8;  // keeps every head 16-byte aligned
  if (hs < (long long)N * D || (H > 1 && bs < hs) || (bs % smul) || (hs % smul))
    return fail(FA_ERR_INVALID_ARG, "%s: bad strides (batch %lld, head %lld)", fn, bs, hs);
  if (khs < (long long)Nk * D || (Hkv > 1 && kbs < khs) || (kbs % smul) || (khs % smul))
    return fail(FA_ERR_INVALID_ARG, "%s: bad key/value strides (batch %lld, head %lld)", fn, kbs, khs);
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)
    return fail(FA_ERR_INVALID_ARG, "%s: tensors must be 16-byte aligned", fn);
  if (!fa::bwd_supported(dtype, D))
    return fail(FA_ERR_UNSUPPORTED, "%s: no kernel for dtype=%s D=%d (f16 / bf16 / fp8_e4m3, D a multiple of 8 up to 128)", fn, fa_dtype_name(dtype), D);
  if ((double)(N > Nk ? N : Nk) * D * 2 >= 4294967296.0) return fail(FA_ERR_INVALID_ARG, "%s: one head exceeds 4 GiB", fn);
  // head dims other than 64 / 128 run on zero-padded rows whose padding is fetched from offset 2^31 + ... (fa_bwd_kernels.hip, PAD)
  if (D != 64 && D != 128 && D != 256 && (double)((N > Nk ? N : Nk) + 128) * D * 2 >= 2147483648.0)  // (+128: rows past the end are addressed too)
    return fail(FA_ERR_INVALID_ARG, "%s: one head exceeds 2 GiB (head dims other than 64 / 128)", fn);
  if (bs < 0 || kbs < 0) return fail(FA_ERR_INVALID_ARG, "%s: negative batch stride", fn);
  if ((long long)B * H > 0x7fffffffLL / (((N > Nk ? N : Nk) + 127) / 128)) return fail(FA_ERR_INVALID_ARG, "%s: grid too large", fn);
  if (dtype == FA_DTYPE_FP8_E4M3 && ((uintptr_t)workspace & 15)) return fail(FA_ERR_INVALID_ARG, "%s: workspace must be 16-byte aligned", fn);
  if (dtype == FA_DTYPE_FP8_E4M3) {
    // e4m3 Q, K, V (O and dO are bf16, as fa_fwd writes O for this dtype): widen them into the workspace behind delta
    // (layout of fa_bwd_workspace_bytes_ex) and run the bf16 kernels on the copies
    char *w = (char *)workspace + align256((long long)B * H * N * 4);
    const long long eq = extent16(B, H, N, D, bs, hs), ek = extent16(B, Hkv, Nk, D, kbs, khs);
    void *q16 = w, *k16 = w + align256(2 * eq), *v16 = w + align256(2 * eq) + align256(2 * ek);
    hipError_t ec = fa::launch_widen_e4m3(q, q16, eq, (hipStream_t)hip_stream);
    if (ec == hipSuccess) ec = fa::launch_widen_e4m3(k, k16, ek, (hipStream_t)hip_stream);
    if (ec == hipSuccess) ec = fa::launch_widen_e4m3(v, v16, ek, (hipStream_t)hip_stream);
    if (ec != hipSuccess) return fail(FA_ERR_LAUNCH, "%s: launch failed: %s", fn, hipGetErrorString(ec));
    q = q16; k = k16; v = v16;
    dtype = FA_DTYPE_BF16;
  }
  hipError_t e = fa::launch_bwd(q, k, v, o, d_o, lse, dq, dk, dv, (float *)workspace, B, H, Hkv, N, Nk, D, scale, bs, hs, kbs, khs,
                                is_causal ? 1 : 0, dtype, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(FA_ERR_LAUNCH, "%s: launch failed: %s", fn, hipGetErrorString(e));
  return FA_OK;
}

int fa_bwd(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse, float *dq,
           float *dk, float *dv, void *workspace, int B, int H, int N, int D, float scale, long long batch_stride,
           long long head_stride, int is_causal, int dtype, void *hip_stream) {
  return bwd_impl("fa_bwd", q, k, v, o, d_o, lse, dq, dk, dv, workspace, B, H, H, N, N, D, scale, batch_stride, head_stride,
                  batch_stride, head_stride, is_causal, dtype, hip_stream);
}

int fa_bwd_ex(const void *q, const void *k, const void *v, const void *o, const void *d_o, const float *lse, float *dq,
              float *dk, float *dv, void *workspace, int B, int Hq, int Hkv, int Nq, int Nk, int D, float scale,
              long long q_batch_stride, long long q_head_stride, long long kv_batch_stride, long long kv_head_stride,
              int is_causal, int dtype, void *hip_stream) {
  return bwd_impl("fa_bwd_ex", q, k, v, o, d_o, lse, dq, dk, dv, workspace, B, Hq, Hkv, Nq, Nk, D, scale, q_batch_stride,
                  q_head_stride, kv_batch_stride, kv_head_stride, is_causal, dtype, hip_stream);
}

}  // extern "C"
