"""Where the time of the elimination sweep goes when a chunk is ONE tile (a window alone on the chip).  Timing experiment: needs the
library built with -DSLSLAM_K1_TIMING=1 (SLSLAM_EXTRA_FLAGS) and SLSLAM_DEBUG_ABLATE set (the stamp buffer is allocated then)."""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "8192"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
names = ["start: chunk, window, state, camera table, zero S", "first tile: descriptor, observations, lines (waited for)", "tiles", "partial: LDS -> memory", "partial: wave sums of the scalars", "partial: scalar stores", "(start: chunk, window, state loads)", "(start: camera table)",
         "backsub start: chunk, window, state, camera tables", "backsub first tile (waited for)", "backsub tiles", "backsub sums + stores"]
for label, kw in (("W=10 house", dict(num_lines=74, num_kf=20, num_free=10, mean_track=16.5)), ("2000 lines", dict(num_lines=2000))):
    w = synth.make_window(5, **kw)
    b = capi.LBABatch(); b.add(w); b.finalize(use_graph=0)
    def read():
        ph = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double))); return ph
    b.solve(); b.download()
    it = b.summary(0)["num_successful_steps"] + b.summary(0)["num_unsuccessful_steps"]
    ph0 = read()
    n = 5
    for _ in range(n): b.reset(); b.solve()
    b.download()
    d = (read() - ph0) / (n * it * (abs(b.window_chunks(0)) % 1000))
    print(json.dumps({"window": label, "chunks": b.window_chunks(0), "cycles_per_chunk_sweep": {nm: round(float(d[i])) for i, nm in enumerate(names) if nm != "-"},
                      "elimination total": round(float(d[:6].sum())), "back-substitution total": round(float(d[8:12].sum()))}))
    b.close()
