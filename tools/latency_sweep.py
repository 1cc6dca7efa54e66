"""Single-window latency (the call protocol of slam.cpp:924-944: one window per keyframe) against the number of chunk
workgroups per window, with per-kernel hipEvent times.  python tools/latency_sweep.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slslam_amd import capi, synth
for lines in (2000, 500):
    w = synth.make_window(5, num_lines=lines)
    for chunks in (0, 25, 49, 97, 194):
        for graph in (1, 0):
            b = capi.LBABatch(); b.add(w); b.finalize(use_graph=graph, chunks_per_window=chunks)
            b.solve(); b.download()
            s = b.summary(0)
            t = time.perf_counter()
            for _ in range(30): b.reset(); b.solve()
            b.download()
            dt = (time.perf_counter() - t) / 30
            out = {"lines": lines, "chunks": chunks, "graph": graph, "ms_per_solve": round(dt * 1e3, 4), "steps": (s["num_successful_steps"], s["num_unsuccessful_steps"])}
            if not graph:
                b.set_profiling(True)
                for _ in range(5): b.reset(); b.solve()
                b.download()
                out["kernel_us_per_launch"] = {k: round(1e3 * v[0] / v[1], 2) for k, v in b.kernel_times().items() if v[1] > 0}
            print(json.dumps(out), flush=True)
            b.close()
