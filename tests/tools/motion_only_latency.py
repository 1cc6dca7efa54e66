"""Developer tool: latency of one motion-only BA call (SLAM::motion_only_ba, once per frame) through slslam_lba_solve."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from slslam_amd import capi, synth
w = synth.make_motion_only(5, num_lines=150)
for f in (0, 1):
    capi.lba_solve(w, lba_fused_motion_only=f)
    t = time.perf_counter()
    for _ in range(50): capi.lba_solve(w, lba_fused_motion_only=f)
    print("one-shot motion-only BA (150 lines), fused=%d: %.3f ms per call" % (f, (time.perf_counter() - t) / 50 * 1e3))
