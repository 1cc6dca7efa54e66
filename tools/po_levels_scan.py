import os, subprocess, sys
code = r'''
import sys
sys.path.insert(0, "/root/repo")
from slslam_amd import capi, synth
for N, loops in ((260, 8), (520, 16), (1000, 30)):
    g = synth.make_pose_graph(7, num_poses=N, num_loops=loops)
    for _ in range(5): capi.po_solve(g)
    best = 1e9
    for _ in range(7):
        x, s, tm = capi.po_solve_timed(g)
        best = min(best, tm["total_ms"])
    st = capi.po_structure(g)
    n1 = st["level1_chains"]
    print("  N=%d: %.3f ms, chains %d + %d, longest L1 %d, upper %s" % (N, best, n1, len(st["chains"]) - n1, max(c[1] for c in st["chains"][:n1]), sorted(set(c[1] for c in st["chains"][n1:]))))
'''
for lv in ["default", "2", "3", "4", "5", "6"]:
    env = dict(os.environ)
    if lv != "default": env["SLSLAM_PO_LEVELS"] = lv
    print("levels", lv); sys.stdout.flush()
    subprocess.call([sys.executable, "-c", code], env=env)
