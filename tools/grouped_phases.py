"""Where the cycles of the grouped matrix-core elimination sweep go (timing experiment: library built with -DSLSLAM_K1_TIMING=1 through
SLSLAM_EXTRA_FLAGS, SLSLAM_DEBUG_ABLATE set so that the stamp buffer exists).  Stamps wait for the wave's LDS operations, so the
phases are what a wave spends in them with its queues drained.   python tools/grouped_phases.py [windows] [elim]"""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "8192"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
elim = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = ["prologue", "linearise (+ tile head)", "line block + scans", "factor", "F rows + records", "prefetch issue", "matrix-core lines", "(loop exit)", "epilogue", "fourth block row"]
ws = [synth.make_window(100 + i, num_lines=2000) for i in range(min(nwin, 16))]
b = capi.LBABatch()
for i in range(nwin): b.add(ws[i % len(ws)])
b.finalize(use_graph=0, lba_elimination=elim)
def read():
    ph = np.zeros(16); capi.lib().slslam_debug_phase_cycles(b._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double))); return ph
b.solve(); b.download()
its = sum(b.summary(i)["num_successful_steps"] + b.summary(i)["num_unsuccessful_steps"] for i in range(nwin))
tiles = sum(b.counts_of(i)["tiles"] if hasattr(b, "counts_of") else 0 for i in range(nwin))
ph = read()
chunks = sum(abs(b.window_chunks(i)) % 1000 for i in range(nwin))
print(json.dumps({"windows": nwin, "elim": elim, "chunks": chunks, "lm_iterations": its,
                  "s_memtime ticks per chunk sweep (100 MHz)": {nm: round(float(ph[i]) / (its / nwin * chunks), 1) for i, nm in enumerate(names)},
                  "share": {nm: round(float(ph[i] / ph[:10].sum()), 3) for i, nm in enumerate(names)}}))
b.close()
