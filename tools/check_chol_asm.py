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
        # the flags of visual_marker_mapping_amd/csrc/Makefile for this file (FLAGS + CHOL_FLAGS)
        cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-mllvm",
               "-amdgpu-mfma-vgpr-form=1", "--offload-device-only", "-S", SRC, "-o", out]
        if stamps:
            cmd.insert(1, "-DVMM_STAMPS")
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       cwd=os.path.dirname(SRC))
        return open(out).read().splitlines()


def audit(lines):
    """Returns (number of asm loads, number of asm waits, list of violations) for k_chol_step."""
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().split(":")[0].startswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end") )
    pending = {}      # register -> (issue sequence number, line number) of the load that will fill it
    state = {"in_asm": False, "seq": 0}
    counts = {"loads": 0, "waits": 0}
    bad = []
    # loop latches: hipcc rotates the tile loop, so the second half of its body (with the final waits) can sit in front
    # of the first half (with the requests) in the text -- a request is then followed to its wait around the back edge
    labels = {}
    for i in range(start + 1, end):
        l = lines[i].strip()
        if l.startswith(".LBB") and ":" in l:
            labels[l.split(":")[0]] = i
    latch = {}
    for i in range(start + 1, end):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", lines[i].strip())
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            latch[i] = labels[m.group(1)]

    def step(i, count):
        l = lines[i].strip()
        if l.startswith(";;#ASMSTART"):
            state["in_asm"] = True
            return
        if l.startswith(";;#ASMEND"):
            state["in_asm"] = False
            return
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            return
        code = l.split(";")[0]
        if state["in_asm"] and code.startswith("global_load_dword"):
            dst = code.split(",")[0]
            state["seq"] += 1
            for r in regs_of(dst):
                if r in pending:
                    bad.append((i + 1, "asm load overwrites v%d still pending from line %d" % (r, pending[r][1])))
                pending[r] = (state["seq"], i + 1)
            src = ",".join(code.split(",")[1:])
            for r in regs_of(src) & set(pending) - regs_of(dst):
                bad.append((i + 1, "asm load addresses through pending v%d" % r))
            counts["loads"] += count
            return
        m_wait = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", code) if state["in_asm"] else None
        if m_wait:
            # loads return in order: vmcnt(N) leaves the N most recently issued ones in flight
            keep = int(m_wait.group(1))
            order = sorted(set(q for q, _ in pending.values()))
            alive = set(order[len(order) - keep:]) if keep else set()
            for r in [r for r, (q, _) in pending.items() if q not in alive]:
                del pending[r]
            counts["waits"] += count
            return
        touched = regs_of(code) & set(pending)
        for r in sorted(touched):
            bad.append((i + 1, "`%s` names v%d between its load (line %d) and the wait" % (code, r, pending[r][1])))

    for i in range(start + 1, end):
        step(i, 1)
        if i in latch and pending and any(latch[i] < ln - 1 <= i for _, ln in pending.values()):
            # requests of this loop still in flight at its latch: once more through the body, up to the first of them
            stop = min(ln for _, ln in pending.values()) - 1
            saved_asm = state["in_asm"]
            state["in_asm"] = False
            for j in range(latch[i], stop):
                step(j, 0)
                if not pending:
                    break
            state["in_asm"] = saved_asm
            if pending:
                bad.append((i + 1, "loads still in flight one loop iteration later: %s" % sorted(pending)))
                pending.clear()
    n_loads, n_waits = counts["loads"], counts["waits"]
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
