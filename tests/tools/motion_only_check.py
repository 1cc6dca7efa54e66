"""Developer tool: the fused one-launch motion-only solve against the general path and the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from slslam_amd import capi, synth
from oracle import pyoracle as O
worst = 0.0
for seed in range(40):
    w = synth.make_motion_only(700 + seed, num_lines=int(20 + 7 * seed), noise_px=[0.0, 0.5, 2.0][seed % 3])
    if seed % 2:            # far from the optimum: rejected steps, radius adaptation
        rng = np.random.default_rng(seed)
        w["parameters"] = w["parameters"].copy()
        w["parameters"][:3] += rng.normal(0, [0.02, 0.08, 0.3][seed % 3], 3)
        w["parameters"][3:6] += rng.normal(0, [0.1, 0.5, 2.0][seed % 3], 3)
    kw = {} if seed % 4 else {"max_num_iterations": seed % 7}
    if seed % 8 == 3: kw["max_num_iterations"] = 40
    if seed % 3 == 1: kw["min_relative_decrease"] = [1.5, 0.999999, 1.0000005][seed % 9 // 3]      # forces rejected steps
    if seed % 5 == 0: kw["huber_delta"] = 0.0
    x0, s0, t0 = capi.lba_solve(w, lba_fused_motion_only=0, **kw)
    x1, s1, t1 = capi.lba_solve(w, lba_fused_motion_only=1, **kw)
    okw = {k: v for k, v in kw.items() if k != "huber_delta"}
    xo, so, to = O.lba_solve(w, huber_delta=kw.get("huber_delta", 1.0 / 406.05), **okw)
    same = (s0["num_successful_steps"], s0["num_unsuccessful_steps"], s0["termination_type"]) == (s1["num_successful_steps"], s1["num_unsuccessful_steps"], s1["termination_type"]) == (so["num_successful_steps"], so["num_unsuccessful_steps"], so["termination_type"])
    dx = max(np.abs(x0 - x1).max(), np.abs(xo - x1).max())
    dc = abs(s1["final_cost"] - so["final_cost"]) / max(so["final_cost"], 1e-300)
    tr = max((abs(a["cost"] - b["cost"]) / max(abs(a["cost"]), 1e-300) for a, b in zip(to, t1)), default=0.0) if len(to) == len(t1) else 1.0
    worst = max(worst, dx)
    flag = "" if same and dx < 1e-6 and dc < 1e-7 and tr < 1e-6 else "  <-- MISMATCH"
    print("seed %2d L=%3d %s steps %d+%d/%d+%d/%d+%d term %d/%d/%d  dx %.1e dcost %.1e dtrace %.1e%s" % (
        seed, w["num_lines"], kw, s0["num_successful_steps"], s0["num_unsuccessful_steps"], s1["num_successful_steps"], s1["num_unsuccessful_steps"],
        so["num_successful_steps"], so["num_unsuccessful_steps"], s0["termination_type"], s1["termination_type"], so["termination_type"], dx, dc, tr, flag))
print("worst dx", worst)
