"""The hand-issued loads of k_chol_step's trailing-update loop (inline-asm global_load + a separate asm s_waitcnt 16
MFMA steps later) are only safe while hipcc schedules nothing that names their destination registers in between.
tools/check_chol_asm.py audits the generated assembly of the normal and the -DVMM_STAMPS build (hipcc cross-compiles
without a GPU; ~35 s)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_an_asm_load_destination_before_its_wait():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_chol_asm.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, universal_newlines=True)
    assert out.returncode == 0, out.stdout
    # 2 call sites x (32 of the rank-64 loop + 48 of the rank-128 loop: C values, this tile's second panel, the next
    # tile's first panel)
    assert "normal build: 160 asm loads" in out.stdout and "stamps build: 160 asm loads" in out.stdout
    assert out.stdout.count("0 violations") == 2
