"""Timeline of one elimination launch: when did every chunk's wave start and end (library built with -DSLSLAM_K1_WALL=1, SLSLAM_DEBUG_ABLATE set so
that the stamp buffer exists)?  Prints the launch's makespan, the mean chunk duration, how busy the 2048 wave slots were and the spread of the
finishing times.   python tools/chunk_timeline.py [windows] [iteration to look at, 1-based, default 6] [chunks per window] [lba_elimination] [elimination | backsub]"""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "8192"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
it = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cpw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
elim = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ws = [synth.make_window(i, num_lines=2000) for i in range(min(nwin, 64))]
b = capi.LBABatch()
for i in range(nwin): b.add(ws[i % len(ws)])
b.finalize(use_graph=0, max_num_iterations=it, chunks_per_window=cpw, lba_elimination=elim)           # the LAST sweep of the solve is the one whose stamps remain
b.solve(); b.download(); b.reset(); b.solve(); b.download()
size = ctypes.c_longlong(0)
capi.lib().slslam_debug_read_cycles(b._h, None, 0, ctypes.byref(size))
raw = np.zeros(size.value, dtype=np.uint64)
capi.lib().slslam_debug_read_cycles(b._h, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), size.value, ctypes.byref(size))
nchunk = sum(abs(b.window_chunks(i)) % 1000 for i in range(nwin))
t = raw[:32 * nchunk].reshape(nchunk, 32)
kernel = sys.argv[5] if len(sys.argv) > 5 else "elimination"
s0, s1 = (30, 31) if kernel == "elimination" else (28, 29)                          # words of the stamp buffer: back-substitution 28 / 29
start, end = t[:, s0].astype(np.float64) * 0.01, t[:, s1].astype(np.float64) * 0.01      # microseconds
ok = end > start
start, end = start[ok], end[ok]
t0 = start.min(); start -= t0; end -= t0
dur = end - start
makespan = end.max()
slots = 2048
print(json.dumps({"kernel": kernel, "windows": nwin, "chunks": int(ok.sum()), "lba_elimination": b.elimination(), "sweep": it, "chunks_per_window": cpw,
                  "makespan_us": round(makespan, 1), "chunk_us": {"mean": round(dur.mean(), 1), "std": round(dur.std(), 1), "min": round(dur.min(), 1), "max": round(dur.max(), 1)},
                  "slot_busy_fraction": round(dur.sum() / (slots * makespan), 3),
                  "ideal_makespan_us_if_perfectly_packed": round(dur.sum() / slots, 1),
                  "finish_time_percentiles_us": {str(p): round(float(np.percentile(end, p)), 1) for p in (50, 90, 99, 100)},
                  "start_time_percentiles_us": {str(p): round(float(np.percentile(start, p)), 1) for p in (1, 33, 34, 66, 67, 99)}}))
b.close()
