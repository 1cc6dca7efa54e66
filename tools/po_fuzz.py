"""Random pose graphs through the structured factorisation (chains on several levels + dense junction block) against the dense factorisation of
the whole matrix: a path with random extra edges (loop closures of any span, hubs, parallel paths), 40 graphs of 20-700 poses.  Same system,
different elimination order: the solved poses must agree to round-off and the LM decisions must be equal.   python tools/po_fuzz.py [graphs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from slslam_amd import capi, synth
from test_gpu_po import _add_edges

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, worst = 0, 0.0
for k in range(count):
    n = int(rng.integers(20, 700))
    g = synth.make_pose_graph(1000 + k, num_poses=n, num_loops=0)
    extra = []
    for _ in range(int(rng.integers(0, 25))):
        a = int(rng.integers(1, n - 2)); span = int(rng.choice([2, 3, 5, 17, 40, 120, n]))
        b = min(n - 1, a + span)
        if b - a > 1: extra.append((a, b))
    if rng.random() < 0.3:                           # a hub
        h = int(rng.integers(1, n - 1))
        extra += [(min(h, int(v)), max(h, int(v))) for v in rng.choice(np.arange(1, n), size=min(6, n - 2), replace=False) if abs(int(v) - h) > 1]
    g = _add_edges(g, extra, rng)
    st = capi.po_structure(g)
    xs, ss, _ = capi.po_solve(g)
    xd, sd, _ = capi.po_solve(g, po_dense_factor=1)
    d = float(np.abs(xs - xd).max())
    worst = max(worst, d)
    same = all(ss[q] == sd[q] for q in ("num_successful_steps", "num_unsuccessful_steps", "termination_type"))
    if d > 1e-9 or not same:
        bad += 1
        print("DIFF graph %d (poses %d, extra edges %d, chains %d + %d, junction unknowns %d): max |dx| %.2e, steps %s / %s" % (
            k, n, len(extra), st["level1_chains"], len(st["chains"]) - st["level1_chains"], st["num_unknowns"] - st["num_chain_unknowns"], d,
            (ss["num_successful_steps"], ss["num_unsuccessful_steps"], ss["termination_type"]), (sd["num_successful_steps"], sd["num_unsuccessful_steps"], sd["termination_type"])))
print("%d random pose graphs, structured (multi-level chains) against dense: %d differ; worst max |dx| %.2e" % (count, bad, worst))
sys.exit(1 if bad else 0)
