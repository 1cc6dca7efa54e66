"""Is a resident batch slower when it is finalized for refills (room in its arrays, pinned image, launches over the whole chunk array)?  And a
batch REFILLED by the device build against the same windows added the ordinary way?  The solve alone (hipGraph replay), ms per solve."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ws = [synth.make_window(i, num_lines=2000) for i in range(B)]


def timed(bt, n=10):
    for _ in range(3):
        bt.reset(); bt.solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bt.reset(); bt.solve()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for name, opt in (("plain", {}), ("refillable (10 % room)", dict(refill_headroom_percent=10)), ("refillable, host packer only", dict(refill_headroom_percent=10, device_build=-1))):
    bt = capi.LBABatch()
    for w in ws:
        bt.add(w)
    bt.finalize(**opt)
    print("%-32s %.3f ms per solve" % (name, timed(bt)))
    if opt.get("refill_headroom_percent") and opt.get("device_build", 0) == 0:
        s2 = capi.WindowSet(ws[37:] + ws[:37], pinned=True)
        bt.refill(s2)
        print("%-32s %.3f ms per solve" % ("  ... refilled on the device", timed(bt)))
        s2.close()
    bt.close()

# does the solve slow down when more batches live beside it (a stream holds `depth` of them)?
import ctypes as C
bts = []
for k in range(3):
    bt = capi.LBABatch()
    for w in ws:
        bt.add(w)
    bt.finalize(refill_headroom_percent=10)
    bts.append(bt)
    print("%d refillable batches alive: batch 0 %.3f ms per solve, the newest %.3f" % (k + 1, timed(bts[0]), timed(bt)))
# ... and when the solve runs on a stream of its own instead of the null stream
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    bt = bts[0]
    for _ in range(3):
        bt.reset(stream=s.cuda_stream); bt.solve(stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bt.reset(stream=s.cuda_stream); bt.solve(stream=s.cuda_stream)
    torch.cuda.synchronize()
    print("batch 0 on a non-null stream: %.3f ms per solve" % (1e3 * (time.perf_counter() - t0) / 10))
# ... alternating between two batches (what a stream does: every solve meets data the previous solve did not touch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    for bt in bts[:2]:
        bt.reset(); bt.solve()
torch.cuda.synchronize()
print("alternating between two batches: %.3f ms per solve" % (1e3 * (time.perf_counter() - t0) / 10))
