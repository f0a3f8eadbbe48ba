"""What the compiler made of the matrix-core kernels (cross-compiled here, no GPU): register budgets, scratch, and the M0 convention of
the LDS-DMA statements. (Rounds 2-3 also audited the MFMA hazard distances of the paired-block kernel's asm-owned registers: that kernel
and its auditor were retired to tools/experiments/ in round 4.)"""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def makefile_flags(stem):
    """CXXFLAGS + FLAGS_<stem> exactly as csrc/Makefile passes them (an audit of other flags audits another kernel: round 3
    measured registers with -fno-slp-vectorize while the Makefile built the backward without it)."""
    import re

    text = open(os.path.join(ROOT, "flash_attention_metal_amd", "csrc", "Makefile")).read()
    cxx = re.search(r"^CXXFLAGS\s*:=\s*(.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    own = re.search(rf"^FLAGS_{stem}\s*:=\s*(.*)$", text, re.M)
    return cxx + (own.group(1).split() if own else [])


def test_backward_kernels_fit_their_occupancy_without_scratch():
    # The backward kernels are compiled for three (head_dim 64) / two (head_dim 128) / one (head_dim 256) workgroups per CU: 168 / 256 / 512 registers.
    # What must hold is that the hot loops do not spill: head_dim 64 not at all, head_dim 128 at most the 32 bytes the dK/dV
    # kernel keeps OUTSIDE its tile loop (per-head constants, reloaded once per query head of the group; seen in the ISA).
    import re

    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = makefile_flags("fa_bwd_kernels")
    assert "-fno-slp-vectorize" in flags  # packed fp32 + shuffles in front of the bf16 packing: 10 % slower (profiles/r03)
    src = os.path.join(ROOT, "flash_attention_metal_amd", "csrc", "fa_bwd_kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], cwd=tmp,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    rows = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                      r.stderr, re.S)
    seen = {}
    for name, vg, ag, scratch, occ in rows:
        if "bwd_" in name:
            seen[name] = (int(vg), int(ag), int(scratch), int(occ))
    # {dq, dkdv} x {f16, bf16} x {64, 128} x {causal, full} x {exact, zero-padded head dim}, + head_dim 256 (exact only)
    assert len(seen) == 40, sorted(seen)
    for name, (vg, ag, scratch, occ) in seen.items():
        if "Li64E" in name:
            assert vg + ag <= 168 and scratch == 0 and occ == 3, (name, vg, ag, scratch, occ)
        elif "Li256E" in name:
            # one workgroup per CU, the whole 512-register file: dK / dV of a wave's 32 keys are 256 accumulator registers, its K~ / V
            # fragments 128 more. The dK/dV kernel fits; the dQ kernel parks up to 116 B per lane (generality, not a tuned path)
            assert vg + ag <= 512 and occ == 1 and scratch <= (128 if "bwd_dq" in name else 0), (name, vg, ag, scratch, occ)
        else:
            assert vg + ag <= 256 and scratch <= 32 and occ == 2, (name, vg, ag, scratch, occ)


def test_forward_kernels_auto_can_dispatch_do_not_spill():
    # Every forward kernel FA_VARIANT_AUTO / fa_fwd_ex can launch, compiled with the Makefile's flags: no scratch, except the
    # documented 16-28 B of the head_dim-256 kernel and the 20 B the causal head_dim-64 16x16x32 kernel keeps in its RARE
    # path (one tuple parked across the row-maximum pass: seen in the ISA between the hot pass and the PV product, behind the branch).
    import re

    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    procs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for stem in ("fa_mfma_kernel", "fa_mfma16_kernel", "fa_fp8_kernel", "fa_fwd_splitkv_kernel"):
            src = os.path.join(ROOT, "flash_attention_metal_amd", "csrc", stem + ".hip")
            procs[stem] = subprocess.Popen([hipcc] + makefile_flags(stem) + ["--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src,
                                            "-o", "/dev/null"], cwd=tmp, stderr=subprocess.PIPE, text=True)
        logs = {stem: p.communicate()[1] for stem, p in procs.items()}
        assert all(p.returncode == 0 for p in procs.values()), {k: v[-1500:] for k, v in logs.items()}
    rows = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", "".join(logs.values()), re.S)
    seen = {n: int(sc) for n, sc in rows if "fwd_" in n}
    assert len(seen) >= 50, len(seen)
    for name, scratch in seen.items():
        if "fwd_mfma16_kernel" in name and "Li64E" in name:
            assert scratch <= 20, (name, scratch)
        elif "Li256E" in name:
            assert scratch <= 32, (name, scratch)
        elif "fwd_mfma_kernel" in name and "Li64ELb0ELb1E" in name and "FP8" not in name:
            # the four-workgroup form of the plain kernel, non-causal: 128 registers and 16 B that are written in the prologue and
            # read back once in front of the tile loop (seen in the ISA), never inside it
            assert scratch <= 16, (name, scratch)
        else:
            assert scratch == 0, (name, scratch)


def test_lds_dma_statements_own_m0():
    # every LDS-DMA inline asm writes M0 and declares only a "memory" clobber (hipcc rejects "m0" in a clobber list with
    # a warning and reserves the register anyway). What keeps that safe is that NO compiler-generated instruction reads M0 between
    # our s_mov and the buffer_load that consumes it, and that nothing else in these kernels depends on M0: audited here on the ISA --
    # every `buffer_load ... lds` is preceded, inside its own asm block, by the s_mov_b32 m0 that belongs to it, and M0 appears
    # nowhere outside such blocks.
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    procs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for stem in ("fa_mfma16_kernel", "fa_fp8_kernel", "fa_bwd_kernels"):
            d = os.path.join(tmp, stem)
            os.mkdir(d)
            src = os.path.join(ROOT, "flash_attention_metal_amd", "csrc", stem + ".hip")
            procs[stem] = (d, subprocess.Popen([hipcc] + makefile_flags(stem) + ["--cuda-device-only", "-S", src, "-o", os.path.join(d, "k.s")],
                                               cwd=d, stderr=subprocess.PIPE, text=True))
        for stem, (d, p) in procs.items():
            err = p.communicate()[1]
            assert p.returncode == 0, err[-1500:]
            lines = open(os.path.join(d, "k.s")).read().splitlines()
            in_asm, own_m0, dma = False, False, 0
            for ln, t in enumerate(lines, 1):
                u = t.strip()
                if u.startswith(";;#ASMSTART"):
                    in_asm, own_m0 = True, False
                elif u.startswith(";;#ASMEND"):
                    in_asm = False
                elif u and not u.startswith((";", ".")):
                    if "m0" in u.replace(",", " ").split():
                        assert in_asm, (stem, ln, u, "M0 touched outside an asm block")
                        if u.startswith("s_mov_b32 m0"):
                            own_m0 = True
                    if u.startswith("buffer_load") and u.endswith(" lds"):
                        assert in_asm and own_m0, (stem, ln, u, "LDS-DMA without its own M0 write in the same statement")
                        dma += 1
            assert dma > 0, stem
