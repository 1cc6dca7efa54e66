"""One-shot solves of a W = 40 window (80 keyframes, 40 free, ~25 k observations) and of a house-sized W = 40 window: the path
for windows beyond the tiled sweeps (lba_big.h), with the host stages of the call.  python tools/big_window_prof.py [repeat]"""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from slslam_amd import capi, synth

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (kf, free, lines, mt) in ((80, 40, 1500, 30.0), (80, 40, 74, 61.0)):
    w = synth.make_window(5, num_lines=lines, num_kf=kf, num_free=free, mean_track=mt)
    capi.lba_solve(w)
    t0 = time.perf_counter()
    for _ in range(rep):
        x, s, t = capi.lba_solve(w)
    dt = (time.perf_counter() - t0) / rep
    x2, s2, _ = capi.lba_solve(w)
    def stages():
        t = [time.perf_counter()]
        b = capi.LBABatch(); b.add(w); t.append(time.perf_counter())
        b.finalize(use_graph=0); t.append(time.perf_counter())
        b.solve(); t.append(time.perf_counter())
        b.download(); t.append(time.perf_counter())
        b.close(); t.append(time.perf_counter())
        return np.diff(t) * 1e3
    stages()
    st = sum(stages() for _ in range(rep)) / rep
    print("window kf=%d free=%d lines=%d obs=%d: %.2f ms per one-shot solve, steps %d+%d, final cost %.9e, reproducible %s" % (
        kf, free, lines, len(w["camera_index"]), 1e3 * dt, s["num_successful_steps"], s["num_unsuccessful_steps"], s["final_cost"], bool(np.array_equal(x, x2))))
    print("    stages: pack %.2f  finalize (lists + upload) %.2f  enqueue %.2f  GPU + download %.2f  close %.2f ms" % tuple(st))
