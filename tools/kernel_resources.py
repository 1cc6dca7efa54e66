#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of a translation unit (no GPU: hipcc cross-compiles gfx950).
   python tools/kernel_resources.py [lba_api.hip] [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "lba_api.hip"
extra = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-c",
                        os.path.join(ROOT, "slslam_amd", "csrc", src), "-o", d + "/x.o", "-Rpass-analysis=kernel-resource-usage"] + extra,
                       capture_output=True, text=True)
txt = r.stderr
if r.returncode:
    print(txt[-3000:]); sys.exit(1)
K = {"v": r"VGPRs", "a": r"AGPRs", "s": r"ScratchSize \[bytes/lane\]", "o": r"Occupancy \[waves/SIMD\]", "l": r"LDS Size \[bytes/block\]"}
for b in txt.split("Function Name: ")[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(slslam::BatchPtrs.*", "", name).replace("void ", "").replace("slslam::", "")
    g = {k: (re.search(p + r": (\d+)", b).group(1) if re.search(p + r": (\d+)", b) else "?") for k, p in K.items()}
    print("%-52s VGPR %4s AGPR %4s scratch %5s occ %2s LDS %s" % (name[:52], g["v"], g["a"], g["s"], g["o"], g["l"]))
