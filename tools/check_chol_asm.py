#!/usr/bin/env python3
"""Build-time audit of the hand-issued loads in k_chol_step (kernels_chol.hip, chol_update_wg).

The trailing-update loop requests its C values and the next tile's operands with `global_load_*` written as inline
asm and waits for them with a separate `s_waitcnt vmcnt(0)` asm statement 16 MFMA steps later.  hipcc does not know
that the destination VGPRs are not valid in between (cdna_hip_programming.md 5.7 item 1): a copy, a spill or any
other use of them that it schedules into that window would read registers the loads have not filled yet -- silently.
This script compiles the file to assembly (device only) and checks, for the normal and the -DVMM_STAMPS build, that
between every asm-issued global_load and the next asm `s_waitcnt vmcnt(0)` NO instruction names one of its
destination registers.

Usage: python tools/check_chol_asm.py [--stamps]   (exit status 0 = clean)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visual_marker_mapping_amd", "csrc", "kernels_chol.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNEL = "_ZN3vmm11k_chol_step"

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def assembly(stamps):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "--offload-device-only",
               "-S", SRC, "-o", out]
        if stamps:
            cmd.insert(1, "-DVMM_STAMPS")
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       cwd=os.path.dirname(SRC))
        return open(out).read().splitlines()


def audit(lines):
    """Returns (number of asm loads, number of asm waits, list of violations) for k_chol_step."""
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().split(":")[0].startswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end") )
    pending = {}      # register -> line number of the load that will fill it
    in_asm = False
    n_loads = n_waits = 0
    bad = []
    for i in range(start + 1, end):
        l = lines[i].strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        code = l.split(";")[0]
        if in_asm and code.startswith("global_load_dword"):
            dst = code.split(",")[0]
            for r in regs_of(dst):
                if r in pending:
                    bad.append((i + 1, "asm load overwrites v%d still pending from line %d" % (r, pending[r])))
                pending[r] = i + 1
            src = ",".join(code.split(",")[1:])
            for r in regs_of(src) & set(pending) - regs_of(dst):
                bad.append((i + 1, "asm load addresses through pending v%d" % r))
            n_loads += 1
            continue
        if in_asm and code.startswith("s_waitcnt") and "vmcnt(0)" in code:
            pending.clear()
            n_waits += 1
            continue
        touched = regs_of(code) & set(pending)
        for r in sorted(touched):
            bad.append((i + 1, "`%s` names v%d between its load (line %d) and the wait" % (code, r, pending[r])))
    if pending:
        bad.append((end, "loads never waited for: %s" % sorted(pending)))
    return n_loads, n_waits, bad


def main():
    variants = [("normal", False), ("stamps", True)] if "--stamps" in sys.argv or len(sys.argv) == 1 else [("normal", False)]
    rc = 0
    for name, stamps in variants:
        n_loads, n_waits, bad = audit(assembly(stamps))
        print("%s build: %d asm loads, %d asm waits, %d violations" % (name, n_loads, n_waits, len(bad)))
        for line, msg in bad[:20]:
            print("  line %d: %s" % (line, msg))
        if bad or n_loads < 32 or n_waits < 1:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
