#!/usr/bin/env bash
# build_ref.sh -- compile the reference's OWN CPU-oracle loops into
# oracle/_ref/libfa_ref_slices.so (TEST INFRASTRUCTURE ONLY).
#
# /root/reference/main.mm cannot be built as a whole here (Objective-C++ over
# Foundation/Metal, main.mm:1-2), and its CPU checks are inline in main().
# Those regions are plain C++17, so this recipe slices them BY LINE NUMBER from
# the file where it lies, wraps them in extern "C" functions that only declare
# the buffers/constants the slices name, and compiles that with g++ -O2.
# The sliced text lives in a temp dir that is deleted; only the .so lands in
# oracle/_ref/ (git-ignored, shipped to the GPU box like our own .so files).
# No reference source is copied into the repository.
#
#   main.mm:3-8      the reference's own #includes
#   main.mm:12-13    D, SCALE                       (N is a parameter here)
#   main.mm:24-30    initRandom
#   main.mm:128-159  non-causal oracle  (names: N D SCALE q_ptr k_ptr v_ptr O_cpu)
#   main.mm:550-578  causal oracle      (names: N_causal D SCALE qc_f kc_f vc_f O_ref)
set -euo pipefail
REF="${FA_REFERENCE_DIR:-/root/reference}/main.mm"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) -- keeping prebuilt $OUT" >&2
  exit 0
fi
# guard: the slices must still be where SURVEY.md says they are
grep -q 'void initRandom' <(sed -n '24p' "$REF")
grep -q 'for (int i = 0; i < N; ++i)' <(sed -n '128p' "$REF")
grep -q 'O_cpu\[i \* D + d\] = num / den' <(sed -n '157p' "$REF")
grep -q 'std::vector<float> O_ref(N_causal \* D)' <(sed -n '550p' "$REF")
grep -q 'O_ref\[i \* D + d\] = val / sum_exp' <(sed -n '576p' "$REF")

TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
{
  sed -n '3,8p' "$REF"
  sed -n '12,13p' "$REF"
  sed -n '24,30p' "$REF"
  cat <<'EOF'
extern "C" int ref_head_dim() { return D; }
extern "C" float ref_scale() { return SCALE; }
extern "C" void ref_init_random(float *data, int size) { initRandom(data, size); }
extern "C" void ref_noncausal(const float *q_in, const float *k_in,
                              const float *v_in, float *out, int N) {
  float *q_ptr = (float *)q_in, *k_ptr = (float *)k_in, *v_ptr = (float *)v_in;
  std::vector<float> O_cpu(N * D);
EOF
  sed -n '128,159p' "$REF"
  cat <<'EOF'
  memcpy(out, O_cpu.data(), sizeof(float) * (size_t)N * D);
}
extern "C" void ref_causal(const float *q_in, const float *k_in,
                           const float *v_in, float *out, int N_causal) {
  float *qc_f = (float *)q_in, *kc_f = (float *)k_in, *vc_f = (float *)v_in;
EOF
  sed -n '550,578p' "$REF"
  cat <<'EOF'
  memcpy(out, O_ref.data(), sizeof(float) * (size_t)N_causal * D);
}
EOF
} > "$TMP/ref_slices.cpp"
mkdir -p "$OUT"
g++ -std=c++17 -O2 -fPIC -shared -o "$OUT/libfa_ref_slices.so" "$TMP/ref_slices.cpp"
echo "built $OUT/libfa_ref_slices.so"
