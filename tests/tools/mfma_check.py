import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from slslam_amd import capi, synth
from oracle import pyoracle as O
for seed, kw in ((1, dict(num_lines=60)), (2, dict(num_lines=200)), (3, dict(num_lines=500)), (4, dict(num_lines=80, num_kf=6, num_free=3)),
                 (6, dict(num_lines=120, num_kf=24, num_free=10, mean_track=30.0)), (7, dict(num_lines=90, num_kf=8, num_free=6, mean_track=2.0)), (8, dict(num_lines=2000))):
    w = synth.make_window(seed, **kw)
    x0, s0, t0 = capi.lba_solve(w, lba_mfma_schur=0)
    x1, s1, t1 = capi.lba_solve(w, lba_mfma_schur=1)
    xo, so, to = O.lba_solve(w, linear_solver=1)
    print(seed, kw, "steps", s0["num_successful_steps"], s0["num_unsuccessful_steps"], "|", s1["num_successful_steps"], s1["num_unsuccessful_steps"], "|", so["num_successful_steps"], so["num_unsuccessful_steps"],
          "final", s0["final_cost"], s1["final_cost"], so["final_cost"], "dx", np.abs(x0 - x1).max(), np.abs(xo - x1).max(),
          "it1 cost", t0[1]["cost"], t1[1]["cost"], to[1]["cost"])
