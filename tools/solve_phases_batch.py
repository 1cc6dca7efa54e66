"""Phase cycles of the reduced solve in a chip-filling batch (four workgroups per CU; SLSLAM_DEBUG_ABLATE=512), per window and launch:
   python tools/solve_phases_batch.py [windows] [elim]"""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "512"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
elim = int(sys.argv[2]) if len(sys.argv) > 2 else 0
names = ["zero+reduce+scalars", "initial/grad checks", "damping", "diag tile", "panel", "forward(start)", "forward", "backward", "candidate poses", "trailing"]
ws = [synth.make_window(100 + i, num_lines=2000) for i in range(16)]
b = capi.LBABatch()
for i in range(nwin): b.add(ws[i % 16])
b.finalize(use_graph=0, lba_elimination=elim)
def read():
    ph = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double))); return ph
b.solve(); b.download()
ph0 = read()
b.reset(); b.solve(); b.download()
d = (read() - ph0) / (nwin * 10)
print(json.dumps({"windows": nwin, "elim": b.elimination(), "cycles_per_solve_kernel_and_window": {nm: round(float(d[i]), 0) for i, nm in enumerate(names)}, "total": round(float(d[:10].sum()), 0)}))
b.close()
