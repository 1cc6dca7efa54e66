"""Where the latency of the reduced camera solve goes for ONE window (timing experiment, SLSLAM_DEBUG_ABLATE=512)."""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "512"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
names = ["zero+reduce+scalars", "initial/grad checks", "damping", "diag tile", "panel", "forward(start)", "forward", "backward", "candidate poses", "trailing"]
for lines in (2000, 500):
    w = synth.make_window(5, num_lines=lines)
    b = capi.LBABatch(); b.add(w); b.finalize(use_graph=0)
    for _ in range(3): b.reset(); b.solve()
    b.download()
    ph0 = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    n = 20
    for _ in range(n): b.reset(); b.solve()
    b.download()
    ph = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    d = (ph - ph0) / (n * 10)
    print(json.dumps({"lines": lines, "cycles_per_solve_kernel": {nm: round(float(d[i]), 0) for i, nm in enumerate(names)}, "total": round(float(d[:10].sum()), 0)}))
    b.close()
