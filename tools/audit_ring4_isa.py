#!/usr/bin/env python3
"""Audit of the four-wave ring kernel's generated code (danspeech_amd/csrc/rnn_persist_ring4.hip): the kernel leans on things the
compiler does not promise, so the build's assembly is checked for them.

  * an "s" operand of the LDS-DMA assembly is a scalar pair (under scalar-register pressure the compiler hands over a VGPR pair)
  * nothing but the kernel's own assembly writes M0 (the second block of a request pair reuses the M0 of the first)
  * no flat load or store, no scratch, no VGPR spill
  * no v_accvgpr copy in a block that holds MFMAs (W_hh is read from AccVGPRs as the MFMAs' A operand)
  * no vector-memory LOAD visible to the compiler in a block between the phase loop's first and last MFMA (the compiler's own wait
    for one is vmcnt(0): the x-projection requests that are meant to stay in flight would be waited for)

    audit_ring4_isa.py [file.s]   compiles with the Makefile's flags (about two minutes) and reports per instantiation
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "danspeech_amd", "csrc", "rnn_persist_ring4.hip")


def main():
    if len(sys.argv) > 1:                       # an assembly file compiled before
        text = open(sys.argv[1]).read()
    else:
      with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ring4.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDSMI_BUILD", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
               "-S", "--cuda-device-only", SRC, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = 0
    kernels = re.findall(r"^(_ZN4dsmi\S*rnn_persist_ring4_kernel\S*?):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M)
    for name, body in kernels:
        short = re.search(r"kernelI(.*?)EEvNS", name).group(1)
        probs = []
        if re.search(r"global_load_lds_dword(x4)? v\d+, v\[", body) and re.search(r"global_load_lds_dwordx4 v\d+, v\[", body):
            probs.append("LDS-DMA with a VGPR base")
        m0 = [l.strip() for l in body.split("\n") if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        if any(not re.match(r"s_mov_b32 m0, s\d+$", l) for l in m0):
            probs.append("M0 touched outside the kernel's assembly: %s" % [l for l in m0 if not re.match(r"s_mov_b32 m0, s\d+$", l)][:2])
        if re.search(r"\bflat_(load|store)", body):
            probs.append("flat memory access")
        if re.search(r"\bscratch_", body):
            probs.append("scratch access")
        blocks, cur = [], []
        for l in body.split("\n"):
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur); cur = []
            else:
                cur.append(l.strip())
        blocks.append(cur)
        for b in blocks:
            if any(x.startswith("v_mfma") for x in b):
                if any("v_accvgpr" in x for x in b):
                    probs.append("v_accvgpr copy beside MFMAs"); break
        for b in blocks:
            if any(x.startswith("v_mfma") for x in b):
                if any(re.match(r"(global|buffer)_load_dword", x) for x in b):
                    probs.append("a compiler-visible vector-memory load beside MFMAs"); break
        at = text.find(".name:", text.find("amdhsa.kernels"))
        at = text.find(name + "\n", at)              # the kernel's metadata entry (fields in alphabetical order around .name)
        vg = re.search(r"\.vgpr_count:\s*(\d+)\s*\n\s*\.vgpr_spill_count:\s*(\d+)", text[at:]) if at > 0 else None
        ag = re.findall(r"\.agpr_count:\s*(\d+)", text[max(at - 3000, 0):at]) if at > 0 else []
        if vg and int(vg.group(2)):
            probs.append("%s VGPRs spilled" % vg.group(2))
        meta, agpr = vg, (ag[-1] if ag else None)
        print("%-28s registers %s%s: %s" % (short, meta.group(1) if meta else "?", " (AccVGPRs %s)" % agpr if agpr else "", "ok" if not probs else "; ".join(probs)))
        bad += bool(probs)
    print("%d instantiations, %d with findings" % (len(kernels), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
