"""One window of tests/tools/soak.py again (same random sequence): per-iteration costs of the oracle and of the device sweeps.
   python tools/soak_window.py <seed> <index> [index ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from slslam_amd import capi, synth
from oracle import pyoracle          # developer tool: the oracle is the checker here
seed = int(sys.argv[1]); want = sorted(int(a) for a in sys.argv[2:])
rng = np.random.default_rng(seed)
for i in range(max(want) + 1):
    kind = rng.integers(0, 10)
    if kind == 0:
        args = ("mo", int(rng.integers(1, 10 ** 6)), int(rng.integers(10, 120)))
    else:
        free = int(rng.integers(2, 21)) if kind < 8 else int(rng.integers(21, 43))
        kf = free + int(rng.integers(0, free + 3)) if kind < 8 else free + int(rng.integers(10, 45))
        lines = int(rng.integers(12, 400)) if kind < 8 else int(rng.integers(20, 90))
        args = ("w", int(rng.integers(1, 10 ** 6)), lines, max(kf, free), free, float(rng.uniform(3.0, max(3.5, 0.8 * max(kf, free)))))
    if i not in want:
        continue
    w = synth.make_motion_only(args[1], num_lines=args[2]) if args[0] == "mo" else synth.make_window(args[1], num_lines=args[2], num_kf=args[3], num_free=args[4], mean_track=args[5])
    x0, s0, t0 = pyoracle.lba_solve(w, linear_solver=1)
    print("window %d: %s  cameras %d (free %d) lines %d observations %d" % (i, args, w["num_cameras"], w["num_free_cameras"], w["num_lines"], len(w["camera_index"])))
    res = {}
    for name, opt in (("elim1", dict(lba_elimination=1)), ("elim4", dict(lba_elimination=4)), ("elim4 chunks=2", dict(lba_elimination=4, chunks_per_window=2)), ("elim1 chunks=2", dict(lba_elimination=1, chunks_per_window=2))):
        x1, s1, t1 = capi.lba_solve(w, lba_fused_motion_only=0, **opt)
        res[name] = (x1, s1, t1)
        print("  %-16s steps %d+%d term %d  final cost rel diff vs oracle %.2e  max|dx| %.2e" % (name, s1["num_successful_steps"], s1["num_unsuccessful_steps"], s1["termination_type"],
              abs(s1["final_cost"] - s0["final_cost"]) / s0["final_cost"], np.abs(x1 - x0).max()))
        print("      per-iteration cost rel diff: " + " ".join("%.1e" % (abs(a["cost"] - b["cost"]) / abs(b["cost"])) for a, b in zip(t1, t0)))
        print("      accepted:                    " + " ".join("%7d" % a["step_is_successful"] for a in t1))
    print("  oracle           steps %d+%d term %d cost %.6e -> %.6e; radius %s" % (s0["num_successful_steps"], s0["num_unsuccessful_steps"], s0["termination_type"], s0["initial_cost"], s0["final_cost"],
          " ".join("%.1e" % a["trust_region_radius"] for a in t0)))
