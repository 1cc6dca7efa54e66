"""A chip-filling batch (graded chunk sizes, grouped sweep, dispatch order by class) against its windows solved alone with the cut the batch
reports: bytes of the parameters, summaries and iteration traces of `picks` windows.   python tools/batch_vs_solo.py [windows] [picks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
picks = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ws = [synth.make_window(5000 + i, num_lines=2000) for i in range(nwin)]
b = capi.LBABatch()
for w in ws: b.add(w)
b.finalize()
b.solve(); b.download()
elim = b.elimination()
rng = np.random.default_rng(3)
idx = sorted(set([0, nwin - 1] + [int(i) for i in rng.integers(0, nwin, size=picks)]))
bad = 0
cuts = {}
for i in idx:
    cut = b.window_chunks(i); cuts[cut] = cuts.get(cut, 0) + 1
    a = capi.LBABatch(); a.add(ws[i]); a.finalize(chunks_per_window=cut, lba_elimination=elim); a.solve(); a.download()
    same = np.array_equal(a.parameters(0), b.parameters(i)) and a.summary(0) == b.summary(i) and a.trace(0) == b.trace(i)
    if not same:
        bad += 1; print("window %d differs: max |dx| %.3e" % (i, np.abs(a.parameters(0) - b.parameters(i)).max()))
    a.close()
print("batch of %d windows (sweep %d, cuts %s): %d of %d picked windows identical to their solo solves (parameters, summary, trace), %d differ" % (nwin, elim, cuts, len(idx) - bad, len(idx), bad))
b.close()
