"""The block-id divisors of the matrix-core kernels (csrc/fa_common.h: make_fastdiv / FastDiv, used by map_block and
head_bases in place of five generic integer divisions per workgroup) must be EXACT for every 32-bit numerator: a wrong
quotient sends a workgroup to the wrong (batch, head, q block). Host-side check of the same header the kernels include."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include "%s"
using namespace fa;
static unsigned fdiv_h(unsigned n, const FastDiv &f) {  // the device function fdiv() of fa_mfma_common.h, on the host
  const unsigned t = (unsigned)(((unsigned long long)f.mul * n) >> 32);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
int main() {
  unsigned long long bad = 0, cnt = 0;
  const unsigned ds[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 31, 32, 33, 63, 64, 96, 100, 127, 128, 255, 256, 1000, 1024,
                         2047, 2048, 4095, 65535, 65536, 1000003, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
  for (unsigned d : ds) {
    const FastDiv f = make_fastdiv(d);
    for (unsigned long long n = 0; n < (1ull << 32); n += 65521) { ++cnt; bad += fdiv_h((unsigned)n, f) != (unsigned)n / d; }
    for (unsigned k = 0; k < 4096; ++k) {
      const unsigned ns[] = {0xffffffffu - k, k * d + (d - 1), k * d};
      for (unsigned n : ns) { ++cnt; bad += fdiv_h(n, f) != n / d; }
    }
  }
  for (unsigned d = 1; d < 3000; ++d) {
    const FastDiv f = make_fastdiv(d);
    for (unsigned n = 0; n < 100000; n += 7) { ++cnt; bad += fdiv_h(n, f) != n / d; }
  }
  printf("checked %%llu bad %%llu\n", cnt, bad);
  return bad != 0;
}
'''


def test_fastdiv_is_exact():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    hdr = os.path.join(ROOT, "flash_attention_metal_amd", "csrc", "fa_common.h")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "fd.cpp")
        open(src, "w").write(SRC % hdr)
        exe = os.path.join(tmp, "fd")
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", src, "-o", exe], stderr=subprocess.DEVNULL)
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        assert "bad 0" in out.stdout
