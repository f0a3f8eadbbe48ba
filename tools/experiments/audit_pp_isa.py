#!/usr/bin/env python3
"""Hazard audit of the paired-block kernel's ISA (fa_fwd_pp_kernel.hip).

That kernel issues its MFMAs from inline asm, and hipcc pads no hazards around an asm statement
(cdna_hip_programming.md section 5.7): the result of a v_mfma_f32_32x32x16 needs 12 wait states
before anything but the next MFMA that accumulates into exactly the same tuple may touch it -- a
VALU read, a compiler-made copy or spill (v_mov / v_accvgpr_write of a score register), an LDS or
memory instruction. The kernel keeps that distance by construction (the order of its work lists, an
explicit fence on the cold paths); this script checks the ASSEMBLY the compiler actually produced:

    for every asm MFMA, every later instruction that names one of its destination registers must be
    at least MIN_STATES wait states away (an instruction = 1 state, s_nop N = N+1), along every
    forward path (branches propagate the pending state to their target label; loops are iterated
    to a fixed point).

It also checks that nothing but the kernel's own asm names the accumulation registers the kernel owns
(audit_owned_agprs) and reports each kernel's scratch size.

usage: audit_pp_isa.py file.s [--min-states 12]     exit code 1 and a listing if anything is too close.
"""
from __future__ import annotations

import argparse
import re
import sys

REG_RE = re.compile(r"\b([va])(?:(\d+)\b|\[(\d+):(\d+)\])")
LABEL_RE = re.compile(r"^(\.L[A-Za-z0-9_$.]+):")
BRANCH_RE = re.compile(r"^\s*s_(?:c)?branch\w*\s+(\.L[A-Za-z0-9_$.]+)")


def regs_of(text: str):
    out = set()
    for m in REG_RE.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            for i in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((kind, i))
    return out


def operands(instr: str):
    body = instr.split(None, 1)
    return [o.strip() for o in body[1].split(",")] if len(body) > 1 else []


def audit_function(name: str, lines, min_states: int):
    """lines: list of (lineno, text) of one kernel. Returns list of violations."""
    instrs = []  # (lineno, text, label or None)
    for ln, t in lines:
        t = t.split(";")[0].rstrip() if not t.lstrip().startswith(";") else ""
        if not t.strip():
            continue
        m = LABEL_RE.match(t.strip())
        if m:
            instrs.append((ln, None, m.group(1)))
            continue
        if t.strip().startswith("."):
            continue
        instrs.append((ln, t.strip(), None))
    label_state = {}  # label -> {reg: (states_elapsed, mfma_lineno)}
    violations = {}

    def merge(dst, src):
        changed = False
        for r, (e, ln) in src.items():
            if r not in dst or dst[r][0] > e:
                dst[r] = (e, ln)
                changed = True
        return changed

    for _ in range(4):  # fixed point over back edges
        pending = {}
        changed_any = False
        for ln, text, label in instrs:
            if label is not None:
                st = label_state.setdefault(label, {})
                merge(pending, st)
                continue
            mb = BRANCH_RE.match(text)
            op = text.split()[0]
            cost = 1
            if op == "s_nop":
                cost = int(text.split()[1], 0) + 1
            is_mfma = op.startswith("v_mfma")
            touched = regs_of(text)
            if is_mfma:
                ops = operands(text)
                dst = regs_of(ops[0])
                srcc = regs_of(ops[3]) if len(ops) > 3 else set()
                accumulate = dst == srcc
                others = regs_of(ops[1]) | regs_of(ops[2])
                check = (others | (set() if accumulate else (dst | srcc)))
                # an accumulate into exactly the same tuple may follow at once; partial overlaps may not
                if accumulate:
                    check |= {r for r in pending if r in dst and False}
            else:
                check = touched
            for r in check:
                if r in pending and pending[r][0] < min_states:
                    violations[(pending[r][1], ln)] = (r, pending[r][0], text)
            # advance time
            for r in list(pending):
                e, l0 = pending[r]
                e += cost
                if e >= min_states + 8:
                    del pending[r]
                else:
                    pending[r] = (e, l0)
            if is_mfma:
                for r in dst:
                    pending[r] = (0, ln)
            if mb:
                st = label_state.setdefault(mb.group(1), {})
                if merge(st, pending):
                    changed_any = True
                if op == "s_branch":
                    pending = {}
        if not changed_any:
            break
    return [(name, a, b, r, e, t) for (a, b), (r, e, t) in sorted(violations.items())]


ASM_START, ASM_END = ";;#ASMSTART", ";;#ASMEND"
AREG_RE = re.compile(r"\ba(?:(\d+)\b|\[(\d+):(\d+)\])")


def audit_owned_agprs(name: str, lines):
    """The kernel OWNS the accumulation registers a[0 : NACC): only its inline asm may name them (O^T, the Q fragments, the
    staging chunks live there across the whole loop; hipcc knows them only as clobbers). NACC = highest accumulation
    register any asm block names + 1. Any instruction OUTSIDE ;;#ASMSTART/;;#ASMEND that names a register below NACC is
    the compiler parking a value of its own there (a spill, a copy): wrong O, silently. Returns (NACC, violations)."""
    inside = False
    nacc = 0
    outside = []  # (lineno, text, lowest a register named)
    for ln, t in lines:
        st = t.strip()
        if st.startswith(ASM_START):
            inside = True
            continue
        if st.startswith(ASM_END):
            inside = False
            continue
        code = st.split(";")[0]
        if not code or code.startswith("."):
            continue
        regs = []
        for m in AREG_RE.finditer(code):
            regs += [int(m.group(1))] if m.group(1) is not None else list(range(int(m.group(2)), int(m.group(3)) + 1))
        if not regs:
            continue
        if inside:
            nacc = max(nacc, max(regs) + 1)
        else:
            outside.append((ln, code, min(regs)))
    return nacc, [(name, ln, code, lo) for ln, code, lo in outside if lo < nacc]


def split_kernels(path: str, match: str):
    """{mangled name: ([(lineno, text)], scratch bytes per lane)} of the kernels whose name contains `match`."""
    funcs, scratch = {}, {}
    cur = last = None
    for ln, t in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", t)
        if m:
            cur = m.group(1) if match in m.group(1) else None
            last = cur
            if cur:
                funcs[cur] = []
            continue
        if cur:
            funcs[cur].append((ln, t))
            if "s_endpgm" in t:
                cur = None
        elif last and "; ScratchSize:" in t:
            scratch[last] = int(t.split(":")[1])
            last = None
    return {k: (v, scratch.get(k, -1)) for k, v in funcs.items()}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--min-states", type=int, default=12)
    ap.add_argument("--match", default="fwd_pp_kernel")
    a = ap.parse_args()
    funcs = split_kernels(a.asm, a.match)
    bad, owned = [], []
    for name, (lines, scratch) in funcs.items():
        bad += audit_function(name, lines, a.min_states)
        nacc, v = audit_owned_agprs(name, lines)
        owned += v
        print(f"  {name[:70]}: owns a[0:{nacc}), scratch {scratch} B/lane")
    print(f"audited {len(funcs)} kernels, {len(bad)} accesses closer than {a.min_states} wait states to an MFMA result, "
          f"{len(owned)} compiler-made references to asm-owned accumulation registers")
    for name, l0, l1, r, e, t in bad[:40]:
        print(f"  {name[:60]}: MFMA at line {l0} -> line {l1} after {e} states touches {r[0]}{r[1]}: {t}")
    for name, ln, code, lo in owned[:40]:
        print(f"  {name[:60]}: line {ln} names a{lo} outside the kernel's asm: {code}")
    bad += owned
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
