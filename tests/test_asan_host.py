"""The host-only code of libdsmi.so under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not
available on the pool): the ARPA and KenLM-binary readers against a few hundred damaged files, the shard plan and the phrase
gate against their definitions.  `make -C danspeech_amd/csrc asan` builds tools/asan/host_fuzz.cpp with the product's own
sources (lm.cpp.inc, lm_klm.cpp.inc, host_logic.h).  A sanitizer report or a crash fails the test; a refused file is fine."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from danspeech_amd import synthetic as syn
from oracle import klm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "danspeech_amd", "csrc", "build", "host_fuzz_asan")


@pytest.fixture(scope="module")
def exe():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "danspeech_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    return EXE


def _run(exe, args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return r.stdout


def test_shard_plans_and_phrase_gates(exe):
    assert "ok" in _run(exe, ["plan", "7", "400"])


def test_language_model_readers_on_damaged_files(exe, tmp_path):
    arpa = str(tmp_path / "lm.arpa")
    syn.make_arpa(arpa, order=3, n_words=120, seed=3, ngrams_per_order=300)
    good = {"arpa": arpa}
    for mt, name, kw in ((klm.PROBING, "probing", {}), (klm.TRIE, "trie", {}), (klm.QUANT_TRIE, "quant", dict(quant_bits=(6, 5))),
                         (klm.ARRAY_TRIE, "array", dict(array_bits=4)), (klm.QUANT_ARRAY_TRIE, "qa", dict(quant_bits=(8, 8), array_bits=255))):
        p = str(tmp_path / ("lm_%s.klm" % name))
        klm.write_klm(arpa, p, mt, **kw)
        good[name] = p
    assert _run(exe, ["lm"] + list(good.values())).startswith("loaded %d refused 0" % len(good))
    rng = np.random.default_rng(11)
    files = []
    for name, path in good.items():
        blob = bytearray(open(path, "rb").read())
        for k in range(60):
            b = bytearray(blob)
            kind = k % 5
            if kind == 0:                                  # a few flipped bytes anywhere
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
            elif kind == 1:                                # damage in the header / counts / first tables
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, min(len(b), 400)))] = int(rng.integers(0, 256))
            elif kind == 2:                                # truncation
                b = b[:int(rng.integers(0, len(b)))]
            elif kind == 3:                                # a run of 0xff (huge counts, pointers, bit fields)
                at = int(rng.integers(0, len(b)))
                b[at:at + int(rng.integers(1, 24))] = b"\xff" * min(int(rng.integers(1, 24)), len(b) - at)
            else:                                          # garbage appended / a slice repeated
                at = int(rng.integers(0, len(b)))
                b = b[:at] + b[at:at + 64] + b[at:]
            p = str(tmp_path / ("mut_%s_%02d" % (name, k)))
            open(p, "wb").write(bytes(b))
            files.append(p)
    out = _run(exe, ["lm"] + files)
    loaded, refused = int(out.split()[1]), int(out.split()[3])
    assert loaded + refused == len(files) and refused > len(files) // 4
