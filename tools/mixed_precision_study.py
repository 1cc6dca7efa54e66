"""lba_precision = 1 against the double path and the oracle, window by window: step counts, where the accept / reject sequences part, costs per
iteration, final cost, poses, line closest points (what tests/test_gpu_lba.py::test_mixed_precision_solves states as its tolerance)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402  (checker)


def main():
    shapes = [(i, 2000) for i in range(0, 12)] + [(100 + i, 500) for i in range(8)] + [(200 + i, 150) for i in range(8)]
    rows = []
    for seed, n in shapes:
        w = synth.make_window(seed, num_lines=n)
        xd, sd, td = capi.lba_solve(w, lba_elimination=4)
        xm, sm, tm = capi.lba_solve(w, lba_precision=1)
        nc = 6 * w["num_cameras"]
        dec_d = [r["step_is_successful"] for r in td]
        dec_m = [r["step_is_successful"] for r in tm]
        first_diff = next((i for i, (a, b) in enumerate(zip(dec_d, dec_m)) if a != b), None)
        upto = min(len(td), len(tm)) if first_diff is None else first_diff
        iter_cost = max([abs(a["cost"] - b["cost"]) / abs(a["cost"]) for a, b in list(zip(td, tm))[:upto]] or [0.0])
        cpm = np.array([synth.orth_to_av(u)[:3] for u in xm[nc:].reshape(-1, 4)])
        cpd = np.array([synth.orth_to_av(u)[:3] for u in xd[nc:].reshape(-1, 4)])
        near = np.linalg.norm(cpd, axis=1) < 10.0
        rows.append(dict(seed=seed, lines=n, steps_double=[sd["num_successful_steps"], sd["num_unsuccessful_steps"]],
                         steps_mixed=[sm["num_successful_steps"], sm["num_unsuccessful_steps"]], first_different_decision=first_diff,
                         iter_cost_rel_until_then=iter_cost, final_cost_rel=abs(sm["final_cost"] - sd["final_cost"]) / sd["final_cost"],
                         cam=float(np.abs(xm[:nc] - xd[:nc]).max()), cp_near=float(np.abs(cpm - cpd)[near].max()),
                         cp_near_p99=float(np.percentile(np.abs(cpm - cpd)[near].max(axis=1), 99)), cp_near_median=float(np.median(np.abs(cpm - cpd)[near].max(axis=1))),
                         lines_beyond_1mm=int((np.abs(cpm - cpd)[near].max(axis=1) > 1e-3).sum()), lines_near=int(near.sum()),
                         rho_at_split=(td[first_diff]["relative_decrease"], tm[first_diff]["relative_decrease"]) if first_diff is not None else None))
        print(json.dumps(rows[-1]))
    print("windows with identical decisions: %d of %d" % (sum(r["first_different_decision"] is None for r in rows), len(rows)))
    print("lines within 10 m whose closest point moved by more than 1 mm: %d of %d" % (sum(r["lines_beyond_1mm"] for r in rows), sum(r["lines_near"] for r in rows)))
    for k in ("iter_cost_rel_until_then", "final_cost_rel", "cam", "cp_near", "cp_near_p99", "cp_near_median"):
        print("worst %-26s %.3e   median %.3e" % (k, max(r[k] for r in rows), float(np.median([r[k] for r in rows]))))


if __name__ == "__main__":
    main()
