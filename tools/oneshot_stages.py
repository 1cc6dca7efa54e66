"""Where the time of a one-shot slslam_lba_solve goes for windows of the reference's study sizes (house scene: 74 lines, W = 5 / 10 / 20 / 40):
host stages (pack, build + upload, enqueue, GPU + download) of the batch API, the one-shot call, and the resident graph replay."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth

for (kf, free, mt, label) in ((10, 5, 8.4, "W=5"), (20, 10, 16.5, "W=10"), (40, 20, 32.0, "W=20"), (80, 40, 61.0, "W=40")):
    w = synth.make_window(5, num_lines=74, num_kf=kf, num_free=free, mean_track=mt)
    def stages():
        t = [time.perf_counter()]
        b = capi.LBABatch(); b.add(w); t.append(time.perf_counter())
        b.finalize(use_graph=0); t.append(time.perf_counter())
        b.solve(); t.append(time.perf_counter())
        b.download(); t.append(time.perf_counter())
        b.close(); t.append(time.perf_counter())
        return np.diff(t) * 1e3
    stages(); stages()
    st = sum(stages() for _ in range(10)) / 10
    capi.lba_solve(w)
    t0 = time.perf_counter()
    for _ in range(20):
        x, s, _ = capi.lba_solve(w)
    one = 1e3 * (time.perf_counter() - t0) / 20
    b = capi.LBABatch(); b.add(w); b.finalize(use_graph=1)
    b.reset(); b.solve(); b.download()
    t0 = time.perf_counter()
    for _ in range(20):
        b.reset(); b.solve()
    b.download()
    res = 1e3 * (time.perf_counter() - t0) / 20
    path = b.path()
    b.close()
    print("%-5s %4d observations, %d LM iterations: one-shot %.3f ms | resident graph replay %.3f ms (path %d) | stages: pack %.3f build+upload %.3f enqueue %.3f GPU+download %.3f" % (
        label, len(w["camera_index"]), s["num_successful_steps"] + s["num_unsuccessful_steps"], one, res, path, st[0], st[1], st[2], st[3]))
