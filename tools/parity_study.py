"""How close the HIP path and the oracle stay, iteration by iteration and parameter class by parameter class, on
the shapes the GPU parity tests use (tests/test_gpu_lba.py).  The numbers set the tolerances written in those tests;
the oracle's own dense-vs-Schur difference is printed beside them as the conditioning yardstick.

    python tools/parity_study.py > gpurun_out/parity_study.txt      (GPU box; the oracle is the checker here)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth          # noqa: E402
from oracle import pyoracle as oracle       # noqa: E402

SHAPES = [(1, 60, 20, 10, {}), (2, 200, 20, 10, {}), (3, 500, 20, 10, {}), (4, 80, 6, 3, {}), (5, 40, 20, 20, {}),
          (6, 150, 40, 20, {}), (7, 150, 80, 40, {}), (11, 60, 24, 12, dict(mean_track=40.0)),
          (12, 90, 8, 6, dict(mean_track=2.0)), (21, 40, 64, 12, dict(mean_track=300.0)), (1234, 2000, 20, 10, {})]


def rel(a, b, floor=0.0):
    return abs(a - b) / (abs(a) + floor) if (abs(a) + floor) > 0 else 0.0


def classes(w, x0, x1):
    C, L = int(w["num_cameras"]), int(w["num_lines"])
    d = np.abs(x0 - x1)
    cam = d[:6 * C].reshape(C, 6)
    ln = d[6 * C:].reshape(L, 4)
    return dict(cam_rot=float(cam[:, :3].max()), cam_trans=float(cam[:, 3:].max()), line_max=float(ln.max()),
                line_median=float(np.median(ln.max(axis=1))), line_p99=float(np.percentile(ln.max(axis=1), 99)),
                lines_above_1e8=int((ln.max(axis=1) > 1e-8).sum()), worst_line=int(ln.max(axis=1).argmax()),
                worst_line_observations=int((np.asarray(w["line_index"]) == int(ln.max(axis=1).argmax())).sum()))


def main():
    out = []
    for seed, lines, kf, free, kw in SHAPES:
        w = synth.make_window(seed, num_lines=lines, num_kf=kf, num_free=free, **kw)
        xs, ss, ts = oracle.lba_solve(w, linear_solver=1)
        xh, sh, th = capi.lba_solve(w)
        rec = dict(shape=dict(seed=seed, lines=lines, kf=kf, free=free, **kw), steps=[ss["num_successful_steps"], ss["num_unsuccessful_steps"]],
                   same_decisions=[a["step_is_successful"] for a in ts] == [b["step_is_successful"] for b in th],
                   final_cost_rel=rel(ss["final_cost"], sh["final_cost"]), hip_vs_oracle=classes(w, xs, xh), per_iteration=[])
        for a, b in zip(ts, th):
            rec["per_iteration"].append(dict(it=a["iteration"], ok=bool(a["step_is_successful"]), cost=rel(a["cost"], b["cost"]),
                                             radius=rel(a["trust_region_radius"], b["trust_region_radius"]),
                                             step_norm=rel(a["step_norm"], b["step_norm"], 1e-12),
                                             rho=rel(a["relative_decrease"], b["relative_decrease"], 1e-3),
                                             grad=rel(a["gradient_max_norm"], b["gradient_max_norm"], 1e-300)))
        if lines <= 500:
            xd, sd, td = oracle.lba_solve(w, linear_solver=0)           # the oracle's dense normal equations
            rec["oracle_dense_vs_schur"] = classes(w, xs, xd)
            rec["oracle_dense_vs_schur"]["final_cost_rel"] = rel(ss["final_cost"], sd["final_cost"])
            rec["oracle_dense_vs_schur"]["per_iteration_cost"] = [rel(a["cost"], b["cost"]) for a, b in zip(ts, td)]
        out.append(rec)
        print(json.dumps(rec), flush=True)
    worst = {}
    for r in out:
        for pi in r["per_iteration"]:
            for k in ("cost", "radius", "step_norm", "rho"):
                worst.setdefault(pi["it"], {}).setdefault(k, 0.0)
                worst[pi["it"]][k] = max(worst[pi["it"]][k], pi[k])
    print("# worst relative difference per iteration index over all shapes")
    for it in sorted(worst):
        print("#  it %2d  " % it + "  ".join("%s %.2e" % (k, v) for k, v in worst[it].items()))
    for k in ("cam_rot", "cam_trans", "line_max", "line_p99", "line_median"):
        print("# final parameters, worst %-11s hip-vs-oracle %.2e   oracle dense-vs-Schur %.2e" % (
            k, max(r["hip_vs_oracle"][k] for r in out), max(r["oracle_dense_vs_schur"][k] for r in out if "oracle_dense_vs_schur" in r)))
    print("# final cost rel: hip-vs-oracle %.2e" % max(r["final_cost_rel"] for r in out))


if __name__ == "__main__":
    main()
