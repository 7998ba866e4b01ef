"""Opt-in (DSMI_AUDIT=1; two minutes of compile time): the four-wave ring kernel's generated code is what its source assumes
(tools/audit_ring4_isa.py: scalar bases of the LDS-DMA assembly, nothing else writes M0, no v_accvgpr copy beside MFMAs, no
compiler-visible vector-memory load in the phase loop, no scratch).  `make -C danspeech_amd/csrc audit` runs the same."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("DSMI_AUDIT") != "1", reason="set DSMI_AUDIT=1 (compiles the kernel to assembly: about two minutes)")
def test_ring4_generated_code_audit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_ring4_isa.py")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 with findings" in r.stdout
