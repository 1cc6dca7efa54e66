"""Developer tool: where the time of a one-shot slslam_lba_solve goes (pack, build + upload, enqueue, GPU + download)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from slslam_amd import capi, synth
w = synth.make_window(5, num_lines=2000)
def run():
    t = [time.perf_counter()]
    b = capi.LBABatch(); b.add(w); t.append(time.perf_counter())
    b.finalize(use_graph=0); t.append(time.perf_counter())
    b.solve(); t.append(time.perf_counter())
    b.download(); t.append(time.perf_counter())
    x = b.parameters(0); s = b.summary(0); t.append(time.perf_counter())
    b.close(); t.append(time.perf_counter())
    return np.diff(t) * 1e3
run(); run()
acc = sum(run() for _ in range(10)) / 10
print("create+add(pack) %.3f  finalize %.3f  solve(enqueue) %.3f  download(sync) %.3f  get %.3f  close %.3f   total %.3f ms" % (*acc, acc.sum()))
t = time.perf_counter()
for _ in range(10): capi.lba_solve(w)
print("one-shot %.3f ms" % ((time.perf_counter() - t) / 10 * 1e3))
