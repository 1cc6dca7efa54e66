"""Developer tool (lives under tests/ because it uses the oracle as the checker / timed CPU reference): RANSAC trial loops of many frames, one call per frame vs one batched call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slslam_amd import capi, synth
from oracle import pyoracle as O
frames = [synth.make_ransac_pair(100 + i, num_lines=200, noise_px=0.4, outlier_frac=0.3, num_trials=1001) for i in range(64)]
capi.ransac_motion_batch(frames[:2])
t = time.perf_counter(); r1 = [capi.ransac_motion(f["obs0"], f["obs1"], f["lines"], f["samples"]) for f in frames]; t1 = time.perf_counter() - t
t = time.perf_counter(); r2 = capi.ransac_motion_batch(frames); t2 = time.perf_counter() - t
t = time.perf_counter(); r0 = [O.ransac_motion(f["obs0"], f["obs1"], f["lines"], f["samples"]) for f in frames[:8]]; t0 = (time.perf_counter() - t) / 8
same = all(a[:2] == b[:2] and (a[3] == b[3]).all() for a, b in zip(r1, r2))
print("64 frames x 1001 trials x 200 lines: per-frame calls %.2f ms/frame, batched call %.2f ms/frame, oracle (adaptive, sequential) %.2f ms/frame; identical=%s" % (
    1e3 * t1 / 64, 1e3 * t2 / 64, 1e3 * t0, same))
