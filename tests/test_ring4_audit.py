"""The four-wave ring kernel's generated code is what its source assumes (tools/audit_ring4_isa.py: scalar bases of the LDS-DMA
assembly, nothing else writes M0, no v_accvgpr copy beside MFMAs, no compiler-visible vector-memory load in the phase loop, no
scratch).  The Makefile audits the assembly of the very compilation that makes rnn_persist_ring4.o (-save-temps) and keeps the
object only if the audit passes; it leaves the report beside the object.  Here: the report of the object the library was linked
from says so, and the auditor itself still finds what it is there to find."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "danspeech_amd", "csrc", "build")


def test_ring4_object_was_audited_when_it_was_built():
    obj, rep = os.path.join(BUILD, "rnn_persist_ring4.o"), os.path.join(BUILD, "rnn_persist_ring4.audit.txt")
    if not os.path.exists(obj):
        pytest.skip("no object here (a box that received only the linked library)")
    assert os.path.exists(rep), "rnn_persist_ring4.o without its audit report: build it with the Makefile"
    assert os.path.getmtime(rep) <= os.path.getmtime(obj) + 1.0       # the object is moved into place after the audit
    text = open(rep).read()
    assert "instantiations, 0 with findings" in text, text[-1500:]
    n = int(text.strip().splitlines()[-1].split()[0])
    assert n >= 30, n                                                  # every production shape of the kernel was looked at


def test_auditor_flags_a_vgpr_base_and_a_foreign_m0_write(tmp_path):
    """A hand-made kernel body with the two faults the audit exists for."""
    s = tmp_path / "bad.s"
    s.write_text("""
_ZN4dsmi12_GLOBAL__N_124rnn_persist_ring4_kernelILi0ELi13ELi4ELb0ELi0EEEvNS0_9Ring4ArgsE:
.LBB0_1:
\ts_mov_b32 m0, s5
\tglobal_load_lds_dwordx4 v1, v[2:3] sc1
\tglobal_load_lds_dwordx4 v1, v[2:3] offset:1024 sc1
\tv_mfma_f32_16x16x32_f16 v[0:3], a[0:3], v[4:7], v[0:3]
\ts_lshl_b32 m0, s2, 1
.Lfunc_end0:
""")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_ring4_isa.py"), str(s)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1, r.stdout
    assert "VGPR base" in r.stdout and "M0 touched" in r.stdout, r.stdout
