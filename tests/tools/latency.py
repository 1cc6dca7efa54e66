"""Developer tool (lives under tests/ because it uses the oracle as the checker / timed CPU reference): single-window latency (BASELINE configs[1] and [2]) on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slslam_amd import capi, synth
from oracle import pyoracle as O
for lines in (500, 2000):
    w = synth.make_window(5, num_lines=lines)
    capi.lba_solve(w)
    t = time.perf_counter(); n = 5
    for _ in range(n): x, s, _ = capi.lba_solve(w)
    full = (time.perf_counter() - t) / n
    for graph in (0, 1):
        b = capi.LBABatch(); b.add(w); b.finalize(use_graph=graph)
        b.solve(); b.download()
        t = time.perf_counter()
        for _ in range(20): b.reset(); b.solve()
        b.download()
        dt = (time.perf_counter() - t) / 20
        its = s["num_successful_steps"] + s["num_unsuccessful_steps"]
        print("L=%d graph=%d: resident solve %.3f ms (%d LM iterations -> %.0f it/s); host-buffer slslam_lba_solve %.3f ms" % (lines, graph, dt * 1e3, its, its / dt, full * 1e3))
        b.close()
    t = time.perf_counter(); xo, so, _ = O.lba_solve(w, linear_solver=1); dt = time.perf_counter() - t
    print("   oracle 1 thread: %.1f ms (%.0f it/s)" % (dt * 1e3, (so["num_successful_steps"] + so["num_unsuccessful_steps"]) / dt))
g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
capi.po_solve(g)
for f32 in (0, 1):
    t = time.perf_counter()
    for _ in range(3): x, s, _ = capi.po_solve(g, po_factor_fp32=f32)
    print("PO 260 poses fp32=%d: %.2f ms per solve, %d+%d steps" % (f32, (time.perf_counter() - t) / 3 * 1e3, s["num_successful_steps"], s["num_unsuccessful_steps"]))
t = time.perf_counter(); xo, so, _ = O.po_solve(g); print("   PO oracle: %.1f ms" % ((time.perf_counter() - t) * 1e3))
