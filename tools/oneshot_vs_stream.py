"""One window per call, as the reference's SLAM loop calls its solver once per keyframe (src/slam.cpp:924-944): the one-shot slslam_lba_solve against a depth-1
stream of one-window batches (slslam_lba_stream_*: refill + graph replay instead of build + eager launches), windows of different seeds so that every call packs
and uploads new arrays.   python tools/oneshot_vs_stream.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth

for lines, kf, free, mt in ((2000, 20, 10, 9.0), (500, 20, 10, 9.0), (74, 20, 10, 16.5), (74, 10, 5, 8.4)):
    ws = [synth.make_window(100 + i, num_lines=lines, num_kf=kf, num_free=free, mean_track=mt) for i in range(24)]
    for w in ws[:4]: capi.lba_solve(w)
    t0 = time.perf_counter()
    for w in ws: capi.lba_solve(w)
    one = 1e3 * (time.perf_counter() - t0) / len(ws)
    st = capi.LBAStream(depth=1, host_threads=1, refill_headroom_percent=30)
    sets = [capi.WindowSet([w]) for w in ws]
    for s in sets[:4]:
        st.collect(st.submit(s), want_summaries=False)
    t0 = time.perf_counter()
    for s in sets:
        st.collect(st.submit(s), want_summaries=False)
    strm = 1e3 * (time.perf_counter() - t0) / len(ws)
    stats = st.stats(); st.close()
    x, _, _ = capi.lba_solve(ws[5])
    print("%4d lines, %2d + %2d keyframes (%5d observations): one-shot %.3f ms per call | depth-1 stream of one-window batches %.3f ms per call (%d refills, %d builds) | same bytes: %s" % (
        lines, free, kf - free, len(ws[0]["camera_index"]), one, strm, stats["refills"], stats["builds"], bool(np.array_equal(sets[5].parameters(0), x))))
