"""Where the headline batch (1024 x 2000-line windows, default options: grouped matrix-core sweep, graded chunks) stands against the oracle,
window by window and iteration by iteration; beside it the same windows alone through the LDS-atomic sweep (lba_elimination = 1) - the
configuration the TIGHT tolerances of tests/test_gpu_lba.py were measured on (profiles/round2_parity_study.txt)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402  (checker)


def rel(a, b, floor=0.0):
    return abs(a - b) / (abs(a) + floor)


def trace_diffs(t0, t1):
    out = []
    for a, c in zip(t0, t1):
        out.append(dict(it=a["iteration"], same=int(a["step_is_successful"] == c["step_is_successful"]),
                        cost=rel(a["cost"], c["cost"]), radius=rel(a["trust_region_radius"], c["trust_region_radius"]),
                        step=rel(a["step_norm"], c["step_norm"], 1e-12), rho=rel(a["relative_decrease"], c["relative_decrease"], 1e-3)))
    return out


def main():
    B = int(os.environ.get("SLSLAM_HEADLINE_WINDOWS", "1024"))
    npick = int(os.environ.get("SLSLAM_PICKS", "32"))
    ws = [synth.make_window(i, num_lines=2000) for i in range(B)]
    b = capi.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize()
    picks = sorted(set(int(round(x)) for x in np.linspace(0, B - 1, npick)))
    b.solve(); b.download()
    print("elimination", b.elimination(), "cut", b.window_chunks(0))
    rows = []
    for i in picks:
        w = ws[i]
        x0, s0, t0 = pyoracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = b.parameters(i), b.summary(i), b.trace(i)
        x2, s2, t2 = capi.lba_solve(w, lba_elimination=1)
        nc = 6 * w["num_cameras"]
        d1, d2 = trace_diffs(t0, t1), trace_diffs(t0, t2)
        worst1 = {k: max(d[k] for d in d1) for k in ("cost", "radius", "step", "rho")}
        worst2 = {k: max(d[k] for d in d2) for k in ("cost", "radius", "step", "rho")}
        at1 = {k: max(d1, key=lambda d: d[k])["it"] for k in ("cost", "radius", "step", "rho")}
        rows.append(dict(window=i, steps=[s0["num_successful_steps"], s0["num_unsuccessful_steps"]],
                         decisions_equal=[all(d["same"] for d in d1), all(d["same"] for d in d2)],
                         grouped_batch=worst1, grouped_at_iteration=at1, lds_atomic_alone=worst2,
                         cam=[float(np.abs(x0[:nc] - x1[:nc]).max()), float(np.abs(x0[:nc] - x2[:nc]).max())],
                         line=[float(np.abs(x0[nc:] - x1[nc:]).max()), float(np.abs(x0[nc:] - x2[nc:]).max())],
                         final_cost=[rel(s0["final_cost"], s1["final_cost"]), rel(s0["final_cost"], s2["final_cost"])]))
        print(json.dumps(rows[-1]))
    for k in ("cost", "radius", "step", "rho"):
        print("worst %-6s grouped batch %.3e   lds-atomic alone %.3e" % (k, max(r["grouped_batch"][k] for r in rows), max(r["lds_atomic_alone"][k] for r in rows)))
    print("worst cam  %.3e / %.3e   line %.3e / %.3e   final cost %.3e / %.3e" % (
        max(r["cam"][0] for r in rows), max(r["cam"][1] for r in rows), max(r["line"][0] for r in rows), max(r["line"][1] for r in rows),
        max(r["final_cost"][0] for r in rows), max(r["final_cost"][1] for r in rows)))
    b.close()


if __name__ == "__main__":
    main()
