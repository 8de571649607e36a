"""The built library must not read an MFMA result before the matrix pipe has written it (tools/mfma_hazard_lint.py).

gfx950 does not interlock; the compiler pads with s_nop, and clang 22 was seen to under-pad the path that reaches the
first accumulator read through a taken branch (csrc/common.h, drain()).  The symptom was one wrong accumulator row
group on the last tile of every workgroup -- small enough to pass a loose tolerance -- so the built code itself is
checked, on CPU, every time the suite runs.
"""
import os
import subprocess
import sys

from conftest import ROOT


def test_no_mfma_read_after_write_hazard_in_built_library():
    lib = os.path.join(ROOT, "ada-mvs_amd", "libadamvs_hip.so")
    assert os.path.exists(lib), "libadamvs_hip.so is not built: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_lint.py"), lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 hazards" in r.stdout
