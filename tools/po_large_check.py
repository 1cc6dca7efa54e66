import sys, numpy as np
sys.path.insert(0, "/root/repo")
from slslam_amd import capi, synth
for (N, loops) in ((400, 8), (520, 16)):
    g = synth.make_pose_graph(7, num_poses=N, num_loops=loops)
    for name, kw in (("structured", {}), ("dense_fp64", dict(po_dense_factor=1))):
        x, s, tm = capi.po_solve_timed(g, **kw)
        print(N, name, "factor %.3f ms" % tm["factor_ms"], "steps", s["num_successful_steps"], s["num_unsuccessful_steps"], "cost %.6e" % s["final_cost"], "term", s["termination_type"])
