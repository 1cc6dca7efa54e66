"""Developer tool: pose-graph solve, structured vs dense factorisation (BASELINE configs[4] shape)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth
for (N, loops) in ((260, 8), (260, 0), (1000, 30)):
    g = synth.make_pose_graph(7, num_poses=N, num_loops=loops)
    res = {}
    for name, kw in (("structured", {}), ("dense", dict(po_dense_factor=1)), ("dense fp32", dict(po_factor_fp32=1))):
        capi.po_solve(g, **kw)
        t = time.perf_counter()
        for _ in range(5): x, s, tr = capi.po_solve(g, **kw)
        dt = (time.perf_counter() - t) / 5
        res[name] = x
        print("N=%d loops=%d %-11s %.2f ms per solve, %d+%d steps, final cost %.9e" % (
            N, loops, name, dt * 1e3, s["num_successful_steps"], s["num_unsuccessful_steps"], s["final_cost"]))
    print("   max |x_structured - x_dense| = %.3e" % np.abs(res["structured"] - res["dense"]).max())
