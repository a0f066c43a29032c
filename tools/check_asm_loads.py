"""Build-time lint for the inline-asm loads of jlm_gate.hip (cdna_hip_programming.md 5.7 item 1): hipcc does not know
that the destination registers of an `asm volatile("global_load_dwordx4 ...")` are not written until the counted wait, so
a spill, a copy or a reuse of one of them between the load and the `s_waitcnt vmcnt(0)` statement that names it would be
silent corruption.  This script scans the kernel's ISA (hipcc -S) and fails if any compiler-generated instruction touches
a destination register of an asm load before the first asm `s_waitcnt vmcnt(0)` behind it (in text order: conservative).

usage: python tools/check_asm_loads.py [file.s]   (default: compiles jlm_amd/csrc/jlm_gate.hip to a temp file)"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs_of(tok)
    return out


def check(path):
    lines = open(path).read().split("\n")
    bad, kernels = [], 0
    i = 0
    while i < len(lines):
        if lines[i].strip().startswith(".amdhsa_kernel"):
            kernels += 1
        i += 1
    in_asm = False
    pending = {}            # vreg -> line of the asm load
    for n, ln in enumerate(lines, 1):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".Lfunc_end"):
                pending = {}
            continue
        if in_asm:
            m = re.match(r"global_load_dwordx4\s+(v\[\d+:\d+\])", t)
            if m:
                for r in regs_of(m.group(1)):
                    pending[r] = n
            elif t.startswith("s_waitcnt vmcnt(0)"):
                pending = {}
            continue
        if pending:
            hit = all_vregs(t) & set(pending)
            if hit:
                bad.append((n, t, sorted(hit)[:4], pending[sorted(hit)[0]]))
    return bad, kernels


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(prefix="jlm_isa_"), "gate.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                               "--cuda-device-only", "-o", path, os.path.join(REPO, "jlm_amd", "csrc", "jlm_gate.hip")],
                              stderr=subprocess.DEVNULL)
    bad, kernels = check(path)
    txt = open(path).read()
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", txt)
    print("kernels: %d, vgpr spills: %s" % (kernels, spills))
    for n, t, regs, at in bad[:20]:
        print("line %d touches v%s (asm load at line %d): %s" % (n, regs, at, t))
    if bad or any(int(x) for x in spills):
        print("FAIL: %d compiler instructions touch registers of asm loads in flight" % len(bad))
        return 1
    print("ok: no compiler instruction touches an asm load's destination before its wait")
    return 0


if __name__ == "__main__":
    sys.exit(main())
