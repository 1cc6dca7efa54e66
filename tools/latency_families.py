"""Per-kernel-family device time of ONE resident window solve at the reference's study sizes (event-timed launches: the sum is
larger than the graph replay of tools/oneshot_stages.py, the split is what matters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
cases = [("W=5", dict(num_lines=74, num_kf=10, num_free=5, mean_track=8.4)), ("W=10", dict(num_lines=74, num_kf=20, num_free=10, mean_track=16.5)),
         ("W=20", dict(num_lines=74, num_kf=40, num_free=20, mean_track=32.0)), ("W=40", dict(num_lines=74, num_kf=80, num_free=40, mean_track=61.0)),
         ("500 lines", dict(num_lines=500)), ("2000 lines", dict(num_lines=2000))]
for label, kw in cases:
    w = synth.make_window(5, **kw)
    b = capi.LBABatch(); b.add(w); b.finalize(use_graph=0)
    b.solve(); b.download()
    it = b.summary(0)["num_successful_steps"] + b.summary(0)["num_unsuccessful_steps"]
    b.set_profiling(1)
    n = 10
    for _ in range(n): b.reset(); b.solve()
    b.download()
    kt = b.kernel_times()
    print("%-10s n=%3d %2d iterations: " % (label, 6 * kw.get("num_free", 5), it) + "  ".join("%s %.1f us/it (%d launches)" % (k, 1e3 * v[0] / n / it, v[1] // n) for k, v in kt.items() if v[1]))
    b.close()
