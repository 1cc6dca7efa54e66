"""Developer tool: one window set, several batch configurations, kernel-family times.
python tools/sweep.py <windows> <lines> chunks=4,7,14 [steps=3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth  # noqa: E402

nb, lines = int(sys.argv[1]), int(sys.argv[2])
kv = dict(a.split("=") for a in sys.argv[3:])
chunks = [int(x) for x in kv.get("chunks", "0").split(",")]
steps = int(kv.get("steps", 3))
reuse = int(kv.get("reuse", 0))
maxit = int(kv.get("maxit", 10))
ws = [synth.make_window(100 + i, num_lines=lines) for i in range(nb)]
for c in chunks:
    b = capi.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize(use_graph=0, chunks_per_window=c, reuse_elimination=reuse, max_num_iterations=maxit)
    b.reset(); b.solve(); b.download()
    b.set_profiling(True)
    b.iterations(clear=True)
    t = time.perf_counter()
    for _ in range(steps):
        b.reset(); b.solve()
    its = b.iterations()
    dt = time.perf_counter() - t
    b.download()
    kt = b.kernel_times()
    print("chunks=%d  %.0f it/s  %.2f ms/step  " % (c, its / dt, 1e3 * dt / steps) +
          " ".join("%s=%.3f" % (k[:8], v[0] / max(v[1], 1)) for k, v in kt.items() if v[1]), flush=True)
    b.close()
