"""One resident 2000-line window solved repeatedly (hipGraph replay): run under rocprofv3 --kernel-trace to get the per-kernel durations of the latency path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kw = dict(num_kf=int(sys.argv[2]), num_free=int(sys.argv[2]) // 2, mean_track=float(sys.argv[3])) if len(sys.argv) > 3 else {}
w = synth.make_window(5, num_lines=lines, **kw)
b = capi.LBABatch(); b.add(w); b.finalize(use_graph=1)
for _ in range(40):
    b.reset(); b.solve()
b.download(); b.close()
