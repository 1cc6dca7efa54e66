"""Developer experiment: one batch on one stream vs two half-batches on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slslam_amd import capi, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsplit = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
ws = [synth.make_window(100 + i, num_lines=2000) for i in range(nb)]
for ns in nsplit:
    streams = [torch.cuda.Stream() for _ in range(ns)]
    batches = []
    for s in range(ns):
        b = capi.LBABatch()
        for w in ws[s::ns]:
            b.add(w)
        b.finalize(use_graph=1)
        batches.append(b)
    for rep in range(2):
        for b, st in zip(batches, streams):
            b.reset(st.cuda_stream); b.solve(st.cuda_stream)
    torch.cuda.synchronize()
    for b, st in zip(batches, streams): b.iterations(st.cuda_stream, clear=True)
    t = time.perf_counter(); steps = 5
    for _ in range(steps):
        for b, st in zip(batches, streams):
            b.reset(st.cuda_stream); b.solve(st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    its = sum(b.iterations(st.cuda_stream) for b, st in zip(batches, streams))
    print("streams=%d: %.0f it/s, %.2f ms per %d-window step" % (ns, its / dt, 1e3 * dt / steps, nb), flush=True)
    for b in batches: b.close()
