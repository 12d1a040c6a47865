#!/usr/bin/env python3
"""Build-time invariant of the 3M engine (csrc/nls_gemm3m.h): its accumulators live in AGPRs a0..a191 that are
addressed BY NAME in inline asm, so the compiler must (a) allocate all of them to the kernel and (b) never use an
AGPR itself in those kernels.  Compiles the library to ISA and checks both for every kernel that uses the engine.

usage: check_agpr.py [path/to/unity.s]   (without an argument: runs hipcc -S on csrc/nls_unity.hip)
"""
import re, subprocess, sys, tempfile, os

KERNELS = ("k_rotate3", "k_gram3")
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "neo_ls_svm_amd", "csrc")
MINE = (re.compile(r"^\s*v_mfma_f64_16x16x4_f64 a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], a\[\d+:\d+\]"),
        re.compile(r"^\s*v_accvgpr_write_b32 a\[\d+\], 0\s*$"), re.compile(r"^\s*v_accvgpr_read_b32 v\d+, a\[\d+\]\s*$"))
AGPR = re.compile(r"\ba\d+\b|\ba\[\d+(:\d+)?\]")


def function_bodies(text):
    out, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if line.startswith(".Lfunc_end"):
                pass
            if ".end_amdhsa_kernel" in line:
                out[name] = body
                name = None
    return out


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "unity.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm",
                                   "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", os.path.join(CSRC, "nls_unity.hip"), "-o", s],
                                  stderr=subprocess.DEVNULL)
            text = open(s).read()
    bodies = function_bodies(text)
    bad = []
    for k in KERNELS:
        hits = [n for n in bodies if k in n]
        if not hits:
            bad.append(f"{k}: kernel not found")
            continue
        body = bodies[hits[0]]
        nfv = acc = None
        for line in body:
            code = line.split(";")[0]
            m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", code)
            if m: nfv = int(m.group(1))
            m = re.search(r"\.amdhsa_accum_offset (\d+)", code)
            if m: acc = int(m.group(1))
            if code.strip().startswith("."):
                continue
            if AGPR.search(code) and not any(p.match(code) for p in MINE):
                bad.append(f"{k}: compiler-generated AGPR use: {code.strip()}")
        if nfv is None or acc is None or nfv - acc < 192:
            bad.append(f"{k}: kernel descriptor allocates {None if nfv is None else nfv - acc} AGPRs, need 192")
        else:
            print(f"{k}: {acc} arch VGPRs + {nfv - acc} AGPRs, no compiler AGPR use")
    if bad:
        print("\n".join(bad[:20]))
        sys.exit(1)


if __name__ == "__main__":
    main()
