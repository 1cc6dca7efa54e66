"""How sensitive is the simulated run of tools/house_study.py to round-off?  The oracle pipeline at W = 40, sigma = 0.2 px with the
parameters handed to every solve perturbed by 1e-13 (relative): the early windows of a run have no fixed keyframe (gauge-free),
so differences of the last bit grow into different maps.  The spread of the aggregates over such perturbations is the band inside
which two correct implementations of the same solver may differ on this closed-loop run (profiles/round2_house_pipeline_timing.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import house_study as hs          # noqa: E402
from oracle import pyoracle      # noqa: E402  (developer study: the oracle is the subject here)

sigma, W = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2, int(sys.argv[2]) if len(sys.argv) > 2 else 40
base = lambda w, it: pyoracle.lba_solve(w, linear_solver=1, max_num_iterations=it)[:2]


def perturbed(eps, seed):
    rng = np.random.default_rng(seed)

    def f(w, it):
        w2 = dict(w)
        w2["parameters"] = w["parameters"] * (1.0 + eps * rng.standard_normal(len(w["parameters"])))
        return pyoracle.lba_solve(w2, linear_solver=1, max_num_iterations=it)[:2]
    return f


r0 = hs.run(sigma, W, base, frames=400)
print("oracle, sigma %.1f W %d:          iterations / frame %.3f  final cost %.4e  mean position error %.4f m" % (
    sigma, W, r0["avg_iterations"], r0["avg_final_cost"], r0["mean_position_error_m"]))
for s in (1, 2, 3):
    r = hs.run(sigma, W, perturbed(1e-13, s), frames=400)
    print("oracle, inputs x (1 + 1e-13 N(0,1)) #%d: iterations / frame %.3f  final cost %.4e  mean position error %.4f m  max |position - unperturbed| %.3e m" % (
        s, r["avg_iterations"], r["avg_final_cost"], r["mean_position_error_m"], np.abs(r["positions"] - r0["positions"]).max()))
