"""The solve right after a refill against the same solve replayed (reset + solve): device time of the solve alone (events on the stream)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402

B = 1024
ws = [synth.make_window(i, num_lines=2000) for i in range(B)]
bt = capi.LBABatch()
for w in ws:
    bt.add(w)
bt.finalize(refill_headroom_percent=10)
sets = [capi.WindowSet(ws[r:] + ws[:r], pinned=True) for r in (37, 74)]


def ev_time(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for mode in ("device build", "host packer"):
    if mode == "host packer":
        bt.close()
        bt = capi.LBABatch()
        for w in ws:
            bt.add(w)
        bt.finalize(refill_headroom_percent=10, device_build=-1, host_threads=8)
    for k in range(4):
        bt.refill(sets[k % 2]); torch.cuda.synchronize()
        first = ev_time(bt.solve)
        bt.reset(); torch.cuda.synchronize()
        again = ev_time(bt.solve)
        bt.reset(); torch.cuda.synchronize()
        third = ev_time(bt.solve)
        print("%-12s refill %d: first solve %.3f ms, replayed %.3f, %.3f" % (mode, k, first, again, third))

# is it the data or the chip?  A heavy solve of ANOTHER batch right before the first solve of the refilled one: if that one is fast now, what was
# slow was the chip coming out of a light-load state (clocks), not the new data
warm = capi.LBABatch()
for w in ws:
    warm.add(w)
warm.finalize()
for k in range(3):
    bt.refill(sets[k % 2]); torch.cuda.synchronize()
    warm.reset(); warm.solve(); warm.reset(); warm.solve()
    first = ev_time(bt.solve)
    print("after two solves of another batch, refill %d: first solve %.3f ms" % (k, first))
