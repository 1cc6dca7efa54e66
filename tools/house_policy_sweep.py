"""keep the restated LM policy falsifiable.  One command sweeps the items SURVEY.md 8a row 6 marks uncertain - gradient tolerance
absolute or relative to the initial gradient, function_tolerance, tolerance tests before or after the accepted step is applied (re-using the LM diagonal
after a rejected step cannot matter: same Jacobian) - through the re-created house study (tools/house_study.py, CPU oracle) and prints BOTH columns the
reference publishes (average LM iterations and average final cost per counted frame) as ratios ours / file against all 40
matlab_script/result_comp_ancdir_orthonorm/ba_result_orthonorm_err{0.2..1.0}_basize{5,10,20,40}_maxnumiter{10,1000}.txt (read from /root/reference at run
time; nothing of them is committed).  Developer study: the oracle is the subject here.  No GPU.
    python tools/house_policy_sweep.py [--frames 200] [--procs 8] [--out profiles/round5_house_policy_sweep.txt]"""
import argparse
import itertools
import multiprocessing as mp
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/matlab_script/result_comp_ancdir_orthonorm"

VARIANTS = [("restated Ceres 1.7.0 (relative gradient tol., ftol 1e-6, tests before the step)", dict()),
            ("gradient tolerance absolute", dict(policy_variant=1)),
            ("tolerance tests after the accepted step", dict(policy_variant=2)),
            ("both", dict(policy_variant=3)),
            ("function_tolerance 1e-5", dict(function_tolerance=1e-5)),
            ("function_tolerance 1e-4", dict(function_tolerance=1e-4)),
            ("function_tolerance 1e-3", dict(function_tolerance=1e-3)),
            ("function_tolerance 1e-4, tests after the step", dict(function_tolerance=1e-4, policy_variant=2))]


def file_columns(err, basize, maxit):
    txt = open(os.path.join(REF, "ba_result_orthonorm_err%.1f_basize%d_maxnumiter%d.txt" % (err, basize, maxit))).read()
    it = float(re.search(r"Average number of iterations = ([0-9.eE+-]+)", txt).group(1))
    fc = float(re.search(r"Average final costs = ([0-9.eE+-]+)", txt).group(1))
    return it, fc


def one(job):
    vi, err, W, maxit, frames = job
    import house_study as hs
    from oracle import pyoracle
    opt = VARIANTS[vi][1]
    solve = lambda w, it: pyoracle.lba_solve(w, linear_solver=1, max_num_iterations=it, **opt)[:2]
    r = hs.run(err, W, solve, frames=frames, max_iter=maxit)
    return vi, err, W, maxit, r["avg_iterations"], r["avg_final_cost"], r["termination_histogram"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--errs", default="0.2,0.4,0.6,0.8,1.0")
    ap.add_argument("--basizes", default="5,10,20,40")
    ap.add_argument("--maxits", default="10,1000")
    ap.add_argument("--variants", default=",".join(str(i) for i in range(len(VARIANTS))))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    errs = [float(x) for x in a.errs.split(",")]; Ws = [int(x) for x in a.basizes.split(",")]; mis = [int(x) for x in a.maxits.split(",")]
    vs = [int(x) for x in a.variants.split(",")]
    jobs = [(v, e, W, m, a.frames) for v, e, W, m in itertools.product(vs, errs, Ws, mis)]
    jobs.sort(key=lambda j: -j[2] * (3 if j[3] > 10 else 1))                 # long runs first
    with mp.Pool(a.procs) as pool:
        res = pool.map(one, jobs, chunksize=1)
    lines = ["Round 5 - LM policy sweep against the reference's 40 ba_result_orthonorm_* files (tools/house_policy_sweep.py --frames %d; CPU oracle through tools/house_study.py)." % a.frames,
             "ratio = ours / file for the two published columns (iterations per counted frame | final cost per counted frame); a setting 'explains' the files where BOTH are near 1.",
             ""]
    table = {}
    for vi, err, W, maxit, it, fc, th in res:
        fit, ffc = file_columns(err, W, maxit)
        table[(vi, err, W, maxit)] = (it / fit, fc / ffc, it, fit, th)
    for vi in vs:
        lines.append("== %s" % VARIANTS[vi][0])
        for m in mis:
            lines.append("   max_num_iterations %d:   rows sigma (px), columns basize %s: iterations ratio | cost ratio" % (m, " ".join("%d" % W for W in Ws)))
            for e in errs:
                lines.append("     %.1f   " % e + "   ".join("%5.2f | %4.2f" % table[(vi, e, W, m)][:2] for W in Ws))
        ri = np.array([table[(vi, e, W, m)][0] for e in errs for W in Ws for m in mis]); rc = np.array([table[(vi, e, W, m)][1] for e in errs for W in Ws for m in mis])
        sel = [(e, W, m) for e in errs for W in Ws for m in mis if W in (10, 20)]
        ri2 = np.array([table[(vi, e, W, m)][0] for e, W, m in sel]) if sel else ri
        lines.append("   all %d files: iterations ratio median %.2f (min %.2f, max %.2f), within +-10 %%: %d; cost ratio median %.2f (min %.2f, max %.2f)" % (
            len(ri), np.median(ri), ri.min(), ri.max(), int(np.sum(np.abs(ri - 1) <= 0.1)), np.median(rc), rc.min(), rc.max()))
        lines.append("   basize 10 and 20 only (%d files): iterations ratio median %.2f, within +-10 %%: %d" % (len(ri2), np.median(ri2), int(np.sum(np.abs(ri2 - 1) <= 0.1))))
        lines.append("")
    out = "\n".join(lines)
    print(out)
    if a.out:
        open(a.out, "w").write(out + "\n")


if __name__ == "__main__":
    main()
