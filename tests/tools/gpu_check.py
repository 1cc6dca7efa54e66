"""Developer diagnostic (not a test): HIP path vs oracle on a few windows, verbose.
Run on the GPU box:  python tools/gpu_check.py [quick|batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slslam_amd import capi, synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def check_linearise(w):
    b = capi.LBABatch()
    b.add(w)
    b.finalize()
    m = len(w["camera_index"])
    c, r, jc, jl = b.linearise(0, m)
    c0, r0, jc0, jl0 = O.lba_cost(w, w["parameters"], want_jac=True)
    print("linearise: cost %.15g vs %.15g | max|dr| %.3g max|dJc| %.3g max|dJl| %.3g" % (
        c, c0, abs(r - r0).max(), abs(jc - jc0).max(), abs(jl - jl0).max()))
    b.close()


def compare_solve(w, **opt):
    x0, s0, t0 = O.lba_solve(w, linear_solver=1, **{k: v for k, v in opt.items() if k in ("max_num_iterations",)})
    x1, s1, t1 = capi.lba_solve(w, **opt)
    print("oracle :", {k: s0[k] for k in ("num_successful_steps", "num_unsuccessful_steps", "initial_cost", "final_cost", "fixed_cost", "termination_type")})
    print("hip    :", {k: s1[k] for k in ("num_successful_steps", "num_unsuccessful_steps", "initial_cost", "final_cost", "fixed_cost", "termination_type")})
    for a, b in zip(t0, t1):
        print("  it %2d cost %.12e / %.12e  rho %.6f / %.6f  radius %.6e / %.6e  |g| %.4e / %.4e  step %.4e / %.4e  ok %d/%d" % (
            a["iteration"], a["cost"], b["cost"], a["relative_decrease"], b["relative_decrease"],
            a["trust_region_radius"], b["trust_region_radius"], a["gradient_max_norm"], b["gradient_max_norm"],
            a["step_norm"], b["step_norm"], a["step_is_successful"], b["step_is_successful"]))
    print("  len traces", len(t0), len(t1), " max|dx| %.3e" % abs(x0 - x1).max())
    return x0, x1


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    print("devices:", capi.device_count())
    if mode.startswith("batch") and len(sys.argv) > 2:
        pass
    w = synth.make_window(1, num_lines=60)
    check_linearise(w)
    print("--- one iteration")
    compare_solve(w, max_num_iterations=1)
    print("--- ten iterations")
    compare_solve(w)
    print("--- motion only")
    compare_solve(synth.make_motion_only(2, num_lines=40))
    print("--- 500 lines")
    compare_solve(synth.make_window(3, num_lines=500))
    if mode.startswith("batch"):
        nb = int(mode[5:] or 64)
        ws = [synth.make_window(100 + i, num_lines=2000) for i in range(nb)]
        b = capi.LBABatch()
        for w in ws:
            b.add(w)
        b.finalize()
        for rep in range(3):
            b.reset()
            t = time.time()
            b.solve()
            b.download()
            dt = time.time() - t
            its = sum(b.summary(i)["num_successful_steps"] + b.summary(i)["num_unsuccessful_steps"] for i in range(nb))
            print("batch %d windows: %.3f ms, %d iterations -> %.1f it/s" % (nb, dt * 1e3, its, its / dt))
        x0, s0, _ = O.lba_solve(ws[0], linear_solver=1)
        print("window0 max|dx| vs oracle %.3e" % abs(b.parameters(0) - x0).max(), b.summary(0), s0)
        b.set_profiling(True)
        b.reset(); b.solve(); b.download()
        print(b.kernel_times())


if __name__ == "__main__":
    main()
