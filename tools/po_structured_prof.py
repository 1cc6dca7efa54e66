"""Structured pose-graph solves under rocprofv3 --kernel-trace: per-kernel durations of the default path (260 poses, 8 loops)."""
import sys
sys.path.insert(0, "/root/repo")
from slslam_amd import capi, synth
g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
for _ in range(10): capi.po_solve(g)
x, s, tm = capi.po_solve_timed(g)
print("device total %.3f ms, %d+%d steps" % (tm["total_ms"], s["num_successful_steps"], s["num_unsuccessful_steps"]))
