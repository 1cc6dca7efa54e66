"""Workload for the pose-graph profiles (rocprofv3 --kernel-trace / --pmc): the 260-pose / 8-loop-closure graph of
BASELINE config 5, solved three times with the structured factorisation (default) or - `dense` - with the blocked MFMA
Cholesky of the whole 1554 x 1554 system."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth
g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
kw = dict(po_dense_factor=1) if "dense" in sys.argv[1:] else {}
for _ in range(3):
    capi.po_solve(g, **kw)
