#!/usr/bin/env python3
"""Where does a kernel wait for memory right after asking for it?  Compiles lba_api.hip (or the given file) to gfx950 assembly and lists,
per kernel whose name contains the pattern, every s_waitcnt vmcnt(N) with the number of instructions since the N+1-th youngest vector
memory instruction before it (the one the wait is for, if memory returns in order): a small distance is an exposed round trip.
   python tools/isa_waits.py k_eliminate_grouped [file.hip] [max distance to report, default 60]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "lba_api.hip"
maxd = int(sys.argv[3]) if len(sys.argv) > 3 else 60
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "slslam_amd", "csrc", src), "-o", d + "/x.s"], capture_output=True, text=True, check=True)
    txt = open(d + "/x.s").read().split("\n")
name = None
for i, l in enumerate(txt):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        dn = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = dn if pat in dn else None
        if name:
            print("==", name[:100]); vm = []; n = 0
        continue
    if not name:
        continue
    s = l.strip()
    if s.startswith(".Lfunc_end"):
        name = None
        continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        if s.endswith(":") and not s.startswith(";"):
            pass
        continue
    n += 1
    op = s.split()[0]
    if op.startswith(("global_load", "global_store", "global_atomic", "buffer_", "flat_", "scratch_")):
        vm.append((n, s[:70]))
    w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", s)
    if w:
        k = int(w.group(1))
        if len(vm) > k:
            at, what = vm[-1 - k]
            if n - at <= maxd:
                print("  line %6d: %-28s %3d instructions after  %s" % (i + 1, s[:28], n - at, what))
