import sys
sys.path.insert(0, "/root/repo")
from slslam_amd import capi, synth
g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
for _ in range(3): capi.po_solve(g, po_dense_factor=1)
