"""Developer tool (lives under tests/ because it uses the oracle as the checker / timed CPU reference): RANSAC scoring throughput (hypothesis x line evaluations per second)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slslam_amd import capi, synth
from oracle import pyoracle as O
for K, H in ((300, 1000), (2048, 8192)):
    poses, obs, lines, _ = synth.make_ransac_frame(1, num_lines=K, num_hypotheses=H)
    capi.ransac_score(poses, obs, lines)
    t = time.perf_counter(); n = 5
    for _ in range(n): s, m = capi.ransac_score(poses, obs, lines)
    dt = (time.perf_counter() - t) / n
    t = time.perf_counter(); s0, m0 = O.ransac_score(poses[:200], obs, lines); dto = (time.perf_counter() - t) * len(poses) / 200
    print("K=%d H=%d: GPU call (alloc+copies+kernel) %.3f ms = %.2e evals/s; oracle 1 thread %.1f ms = %.2e evals/s; identical=%s" % (
        len(obs), H, dt * 1e3, len(obs) * H / dt, dto * 1e3, len(obs) * H / dto, np.array_equal(s[:200], s0)))
