"""The hazard audit of the paired-block kernel's ISA (tools/audit_pp_isa.py): the checker itself on synthetic
snippets, then the assembly hipcc produces from csrc/fa_fwd_pp_kernel.hip (cross-compiled here, no GPU)."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import audit_pp_isa  # noqa: E402


def run(snippet):
    lines = list(enumerate(snippet.strip().splitlines(), 1))
    return audit_pp_isa.audit_function("k", lines, 12)


def test_checker_flags_a_close_reader_and_accepts_the_legal_forms():
    mf = "\tv_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], a[64:67], 0"
    assert run(mf + "\n\tv_add_f32_e32 v40, v17, v41")  # VALU read 1 state behind the MFMA
    assert run(mf + "\n\tv_accvgpr_write_b32 a130, v31")  # a compiler spill of a score register
    assert run(mf + "\n\tv_mov_b32_e32 v20, v2")  # WAW
    assert not run(mf + "\n\ts_nop 11\n\tv_add_f32_e32 v40, v17, v41")
    assert not run(mf + "\n\tv_mfma_f32_32x32x16_bf16 v[16:31], v[4:7], a[68:71], v[16:31]")  # accumulate chain
    assert run(mf + "\n\tv_mfma_f32_32x32x16_bf16 v[32:47], v[16:19], a[68:71], 0")  # result as A operand: too early
    twelve = "\n".join("\tv_add_f32_e32 v50, v51, v52" for _ in range(12))
    assert not run(mf + "\n" + twelve + "\n\tv_add_f32_e32 v40, v17, v41")
    # a forward branch that skips the padding is a short path
    assert run(mf + "\n\ts_cbranch_vccz .LBB0_1\n" + twelve + "\n.LBB0_1:\n\tv_add_f32_e32 v40, v17, v41")
    # accumulation registers: the asm-owned O tile
    pv = "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]"
    assert run(pv + "\n\tv_accvgpr_read_b32 v9, a3")
    assert not run(pv + "\n\ts_nop 15\n\tv_accvgpr_read_b32 v9, a3")


def test_compiled_kernel_keeps_every_mfma_result_12_states_from_its_first_reader():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "flash_attention_metal_amd", "csrc", "fa_fwd_pp_kernel.hip")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-honor-nans",
                               "-fno-slp-vectorize", "-Wno-division-by-zero", "-DFA_PP_AUDIT_SUBSET", "-save-temps", "-c", src, "-o", "/dev/null"],
                              cwd=tmp, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        assert len(asm) == 1
        os.rename(os.path.join(tmp, asm[0]), os.path.join(tmp, "pp.s"))
        funcs = {}
        cur = None
        import re
        for ln, t in enumerate(open(os.path.join(tmp, "pp.s")), 1):
            m = re.match(r"^(_Z\w+):", t)
            if m:
                cur = m.group(1) if "fwd_pp_kernel" in m.group(1) else None
                if cur:
                    funcs[cur] = []
                continue
            if cur:
                funcs[cur].append((ln, t))
                if "s_endpgm" in t:
                    cur = None
        assert len(funcs) == 4  # bf16: 2 head dims x causal / not (the schedule does not depend on the input type)
        bad = []
        for name, lines in funcs.items():
            assert any("v_mfma" in t for _, t in lines)
            bad += audit_pp_isa.audit_function(name, lines, 12)
        assert not bad, bad[:5]
