"""Two-level chains of the structured pose-graph factorisation: device time of a solve by the piece length of the first level
(SLSLAM_PO_SUBCHAIN; default = sqrt of the longest path), three graphs.   python tools/po_subchain_scan.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
from slslam_amd import capi, synth
for N, loops in ((260, 8), (520, 16), (1000, 30)):
    g = synth.make_pose_graph(7, num_poses=N, num_loops=loops)
    for _ in range(5): capi.po_solve(g)
    best = 1e9
    for _ in range(5):
        x, s, tm = capi.po_solve_timed(g)
        best = min(best, tm["total_ms"])
    st = capi.po_structure(g)
    print("  N=%%d loops=%%d: %%.3f ms device (%%d+%%d steps), chains %%d + %%d, junction unknowns %%d" %% (N, loops, best, s["num_successful_steps"], s["num_unsuccessful_steps"],
          st["level1_chains"], len(st["chains"]) - st["level1_chains"], st["num_unknowns"] - st["num_chain_unknowns"]))
''' % ROOT
for sub in ["auto", "6", "10", "12", "15", "18", "22", "26", "32"]:
    env = dict(os.environ)
    if sub != "auto": env["SLSLAM_PO_SUBCHAIN"] = sub
    else: env.pop("SLSLAM_PO_SUBCHAIN", None)
    print("piece length", sub); sys.stdout.flush()
    subprocess.call([sys.executable, "-c", code], env=env)
