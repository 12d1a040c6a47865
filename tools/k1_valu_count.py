"""VALU instructions per feature in the epilogue of k_featuremap<false, false> (the plane-writing feature map of the fit), from the gfx950 ISA.

The fp64 matrix pipe shares its datapath with the vector ALU (profiles/r02_probe_f64_coexec.log), so K1's floor is not an HBM figure but
    t >= 2 n dk Kf / (fp64 MFMA peak)  +  (VALU instructions per feature) n Kf / (VALU issue rate: 256 CUs x 4 SIMDs x 1 wave64 instruction / 4 cycles)
bench.py reports it as roofline_k1.datapath; this script is where its instruction count comes from.
Usage:  python tools/k1_valu_count.py [nls_unity.s]      (compiles csrc/nls_unity.hip to assembly first when no file is given: ~2 min)"""
import os, re, subprocess, sys, tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(tempfile.gettempdir(), "nls_unity.s")
if len(sys.argv) <= 1:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "include"), "-o", path, os.path.join(ROOT, "neo_ls_svm_amd", "csrc", "nls_unity.hip")], check=True, stderr=subprocess.DEVNULL)
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3nls12k_featuremapILb0ELb0EEEvNS_16FeatureMapParamsE:"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start:end]
last_mfma = max(i for i, l in enumerate(body) if "v_mfma" in l)
blocks, cur = [], None
for l in body[last_mfma + 1:]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        cur = Counter()
        blocks.append(cur)
        continue
    if not t or t.startswith(";") or t.startswith(".") or cur is None:
        continue
    cur[t.split()[0]] += 1
valu = lambda c: sum(v for k, v in c.items() if k.startswith("v_"))
stores = lambda c: sum(v for k, v in c.items() if "global_store" in k)
fast = max(blocks, key=stores)  # the straight-line epilogue of a tile without padded columns: every store of the thread
test = next(b for b in blocks if b.get("v_max3_u32", 0) > 8)  # the one range test per thread (|t| <= 2^30)
feats = stores(fast) / 2  # a cos and a sin plane store per feature
print(f"epilogue block: {valu(fast)} VALU instructions, {stores(fast)} stores -> {feats:.0f} features per thread, {valu(fast) / feats:.2f} VALU per feature")
print(f"range test:     {valu(test)} VALU instructions per thread -> {valu(test) / feats:.2f} per feature")
print(f"total:          {(valu(fast) + valu(test)) / feats:.2f} VALU instructions per feature")
print("mix:", dict(fast.most_common(8)))
