"""Static scan of the library's gfx950 code for three latency anti-patterns that a profiler's per-kernel averages do not show (round 5):

  rmw     load - s_waitcnt vmcnt(0) - store repeated: a read-modify-write written as A[i] -= x entry by entry (the compiler keeps a store
          to A ahead of the next load from A): one memory round trip PER ENTRY
  stage   (loads) - wait - LDS store repeated, or a loop body of one load, one wait, one store: a cooperative copy whose trip count hangs
          on threadIdx.x is not unrolled and pays a round trip per trip
  pred    global loads sitting behind their own scalar branch (per-element bounds tests inside unrolled loops)

Usage:  python tools/isa_scan.py            (compiles csrc/nls_unity.hip to assembly first: ~2 min)
        python tools/isa_scan.py unity.s    (an existing assembly file)
Found and fixed with it in round 5: k_potrf_syrk / k_zpotrf_herk (64 round trips per thread), k_sb_her2k, k_trd_rank2k (16), k_sb_x (8),
k_zpotrf_panel's prologue (32), the copy loops at the head of every small kernel of the band reduction, k_sb_hemm / k_sb_her2k's per-load tests.
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly():
    if len(sys.argv) > 1:
        return sys.argv[1]
    out = os.path.join(tempfile.gettempdir(), "nls_unity.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "neo_ls_svm_amd", "csrc", "nls_unity.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def functions(path):
    fn, lines = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, lines = m.group(1), []
            continue
        if fn is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            yield fn, lines
            fn = None
        elif t and not t.startswith(";"):
            lines.append(t)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    rows = []
    for fn, lines in functions(assembly()):
        seq, pred, nload = [], 0, 0
        for i, t in enumerate(lines):
            if t.startswith(("global_load", "buffer_load")):
                seq.append("L")
                nload += 1
                if any(c.startswith(("s_cbranch_exec", "s_and_saveexec")) for c in lines[max(0, i - 4):i]):
                    pred += 1
            elif t.startswith(("global_store", "buffer_store")):
                seq.append("S")
            elif t.startswith("ds_write"):
                seq.append("D")
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                seq.append("W")
            elif re.match(r"^\.LBB\d+_\d+:", t):
                seq.append("[")
            elif t.startswith("s_cbranch"):
                seq.append("]")
        s = "".join(c for c in seq)
        flat = re.sub(r"[\[\]]", "", s)
        rmw = max([len(r) // 3 for r in re.findall(r"(?:LWS){2,}", flat)] or [0])
        stage = max([r.count("W") for r in re.findall(r"(?:L+WD+){3,}", flat)] or [0])
        loops = len(re.findall(r"\[L{1,2}W[DS]+\]", s))
        if rmw >= 3 or stage >= 3 or loops or pred >= 16:
            rows.append((fn, rmw, stage, loops, pred, nload))
    names = demangle([r[0] for r in rows])
    print(f"{'rmw':>4} {'stage':>5} {'loops':>5} {'pred':>5}/{'loads':<5}  kernel")
    for fn, rmw, stage, loops, pred, nload in sorted(rows, key=lambda r: -(r[1] + r[2] + r[3])):
        n = names.get(fn, fn)
        if "rocprim" in n or "hipcub" in n:
            continue
        print(f"{rmw:4d} {stage:5d} {loops:5d} {pred:5d}/{nload:<5d}  {n[:110]}")


if __name__ == "__main__":
    main()
