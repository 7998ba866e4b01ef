"""CPU-side checks of the C-ABI boundary: libdsmi.so loads, exports every function
include/dsmi.h declares, and the ctypes prototypes cover exactly that set.  No kernels run."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "dsmi.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsmi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_ctypes_prototypes_agree():
    from danspeech_amd import _native
    assert _header_functions() == _native.declared_symbols()


def test_library_exports_every_declared_symbol():
    from danspeech_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.fail("libdsmi.so missing: run `make -C danspeech_amd/csrc`")
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in _header_functions():
        assert hasattr(L, name), name


def test_library_exports_nothing_but_the_abi():
    """-fvisibility=hidden + the linker's export list (csrc/libdsmi.map): the dynamic symbol table holds the functions of
    include/dsmi.h and nothing else -- no C++ helper, no kernel handle."""
    import shutil
    import subprocess
    from danspeech_amd import _native
    if not shutil.which("nm"):
        pytest.skip("no nm")
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _header_functions()


def test_create_rejects_bad_conv_layers_without_gpu():
    """Argument validation happens before any HIP call (reference ConvError, model.py:344-348)."""
    from danspeech_amd import _native
    L = _native.lib()
    d = _native.ModelDesc(conv_layers=0, rnn_type=0, rnn_hidden_size=8, rnn_layers=1, bidirectional=1, context=20,
                          n_labels=33, sample_rate=16000, window_size=0.02)
    assert [f[0] for f in _native.ModelDesc._fields_] == ["conv_layers", "rnn_type", "rnn_hidden_size", "rnn_layers", "bidirectional",
                                                          "context", "n_labels", "sample_rate", "window_size"]      # = dsmi_model_desc
    h = ctypes.c_void_p()
    assert L.dsmi_model_create(ctypes.byref(d), 0, ctypes.byref(h)) == _native.DSMI_ERR_CONV
    assert b"0 convolutional layers" in L.dsmi_last_error(None)
    d.conv_layers = 4
    assert L.dsmi_model_create(ctypes.byref(d), 0, ctypes.byref(h)) == _native.DSMI_ERR_CONV


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from danspeech_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryMissing):
        _native.lib()


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/dsmi.h must compile as C99 on its own (no C++ or HIP types in the signatures)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "dsmi.h"\nint main(void) { dsmi_model_desc d; (void)d; return DSMI_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
