import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from slslam_amd import capi, synth
B = 1024
windows = [synth.make_window(i, num_lines=2000) for i in range(B)]
# resident batch first, as bench.py does
res = {}
if len(sys.argv) > 1 and sys.argv[1] == "resident":
    bt = capi.LBABatch()
    for w in windows: bt.add(w)
    bt.finalize(use_graph=1)
    for _ in range(3):
        bt.reset(); bt.solve()
    bt.download()
    res = {i: bt.parameters(i) for i in (0, 1, B // 2, B - 1)}
    if len(sys.argv) > 2 and sys.argv[2] == "close": bt.close()
import types
# monkeypatch: record per-call times
orig_collect, orig_submit = capi.LBAStream.collect, capi.LBAStream.submit
log = []
def collect(self, t, want_summaries=True):
    a = time.perf_counter(); r = orig_collect(self, t, want_summaries); log.append(("collect", 1e3 * (time.perf_counter() - a))); return r
def submit(self, ws):
    a = time.perf_counter(); r = orig_submit(self, ws); log.append(("submit", 1e3 * (time.perf_counter() - a))); return r
capi.LBAStream.collect, capi.LBAStream.submit = collect, submit
out = bench.streamed_block(windows, 0, 680000.0, res, batches_timed=32, depth=3, host_threads=0, chunks_per_window=0, lba_elimination=0, lba_keep_jacobian=0)
print(json.dumps({k: out[k] for k in ("value", "ms_per_batch", "steady_ms_per_batch", "ms_per_batch_in_submit", "ms_per_batch_waiting_in_collect")}))
print(" ".join("%s%.1f" % (k[0], v) for k, v in log))
