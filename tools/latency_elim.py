"""One resident window (hipGraph replay), ms per 10-iteration solve, per elimination sweep and chunk count:
   python tools/latency_elim.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slslam_amd import capi, synth
import bench
dev = 0
cases = {"2000 lines": synth.make_window(5, num_lines=2000), "500 lines": synth.make_window(5, num_lines=500)}
for W, mt in ((5, 8.4), (10, 16.5), (20, 32.0)):
    cases["study W=%d" % W] = synth.make_window(5, num_lines=74, num_kf=2 * W, num_free=W, mean_track=mt)
for name, w in cases.items():
    if w is None:
        continue
    row = []
    for elim in (1,):
        for ch in (0, 12, 24, 50, 100, 200, 400):
            try:
                _, ms = bench.time_batch([w], dev, 30, 3, lba_elimination=elim, chunks_per_window=ch)
                row.append("elim %d chunks %2d: %.3f" % (elim, ch, ms))
            except Exception as e:
                row.append("elim %d chunks %d: %r" % (elim, ch, e))
    print(name, " | ".join(row))
