"""Where the time of k_big_solve goes for a house-sized W = 40 window (timing experiment, SLSLAM_DEBUG_ABLATE=512):
s_memtime ticks (100 MHz) per launch, by phase."""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "512"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
names = ["load", "-", "panels", "trailing + look-ahead factor", "backward", "(diagonal tiles, inside)"]
w = synth.make_window(5, num_lines=74, num_kf=80, num_free=40, mean_track=61.0)
b = capi.LBABatch(); b.add(w); b.finalize(use_graph=0)
def read():
    ph = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double))); return ph
b.solve(); b.download()
ph0 = read(); it0 = b.summary(0)["num_successful_steps"] + b.summary(0)["num_unsuccessful_steps"]
n = 5
for _ in range(n): b.reset(); b.solve()
b.download()
d = (read() - ph0) / (n * it0)
print(json.dumps({"ticks_per_launch": {nm: round(float(d[i]), 1) for i, nm in enumerate(names)}, "total": round(float(d[:5].sum()), 1), "iterations": it0}))
