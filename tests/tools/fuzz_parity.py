"""Developer tool (lives under tests/ because it uses the oracle as the checker / timed CPU reference): randomised parity sweep of the HIP LBA path against the oracle on many small windows of
varied shape (keyframe counts, free / fixed split, track lengths, noise, robust loss on / off, constant lines,
scrambled observation order).  python tools/fuzz_parity.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slslam_amd import capi, synth  # noqa: E402
from oracle import pyoracle as O     # noqa: E402  (developer tool: the oracle is the checker)



def run(cases, seed=2026, verbose=True, oversize=False, hip_opt=None):
  rng = np.random.default_rng(seed)
  worst = {"cost": 0.0, "x": 0.0, "trace": 0.0}
  bad = 0
  for case in range(cases):
      nkf = int(rng.integers(2, 25))
      nfree = int(rng.integers(2, min(nkf, 20) + 1))
      nl = int(rng.integers(3, 120))
      kw = dict(num_lines=nl, num_kf=nkf, num_free=nfree, noise_px=float(rng.choice([0.0, 0.3, 1.0, 3.0])),
                mean_track=float(rng.choice([2.0, 5.0, 9.0, 30.0])))
      if oversize:                      # windows beyond the tiled sweeps (lba_big.h): > 64 cameras, > 20 free, or lines tracked through > 64 keyframes
          kind = case % 3
          nkf = int(rng.integers(65, 91)) if kind == 0 else int(rng.integers(22, 64))
          nfree = int(rng.integers(21, min(nkf, 45) + 1)) if kind == 1 else int(rng.integers(2, min(nkf, 20) + 1))
          if kind == 2: nkf, nfree = int(rng.integers(66, 80)), int(rng.integers(2, 30))
          nl = int(rng.integers(3, 70))
          kw.update(num_lines=nl, num_kf=nkf, num_free=nfree, mean_track=float(rng.choice([9.0, 30.0, 200.0]) if kind != 2 else 300.0))
      if rng.random() < 0.15:
          kw["line_init"] = "triangulate"
      try:
          if not oversize and rng.random() < 0.12:        # the motion_only_ba shape (one-launch kernel): one free camera, constant lines
              w = synth.make_motion_only(1000 * (seed % 1000) + case, num_lines=nl, noise_px=kw["noise_px"])
              w["parameters"] = w["parameters"].copy()
              w["parameters"][:6] += rng.normal(0, 1.0, 6) * float(rng.choice([0.0, 0.01, 0.2]))
          else:
              w = synth.make_window(1000 * (seed % 1000) + case, **kw)
      except Exception as e:            # generator cannot build this shape (e.g. too few visible lines)
          continue
      m = len(w["camera_index"])
      if rng.random() < 0.5:             # the reference's packer order is by line; any order must work
          perm = rng.permutation(m)
          for k in ("camera_index", "line_index", "observations"):
              w[k] = w[k][perm]
          w["fixed_index"] = w["fixed_index"].reshape(-1, 2)[perm].reshape(-1)
      if rng.random() < 0.2:             # some lines constant (motion-only style)
          const = rng.random(w["num_lines"]) < 0.3
          fi = w["fixed_index"].reshape(-1, 2).copy()
          fi[:, 1] = const[w["line_index"]]
          w["fixed_index"] = fi.reshape(-1)
      opt = {}
      if rng.random() < 0.25:
          opt["huber_delta"] = 0.0
      if rng.random() < 0.2:
          opt["max_num_iterations"] = int(rng.integers(0, 30))
      oo = {k: v for k, v in opt.items() if k != "huber_delta"}
      x0, s0, t0 = O.lba_solve(w, huber_delta=opt.get("huber_delta", 1.0 / 406.05), linear_solver=1, **oo)
      x1, s1, t1 = capi.lba_solve(w, **opt, **(hip_opt or {}))
      ok = True
      n = min(len(t0), len(t1))
      dtr = max((abs(a["cost"] - b["cost"]) / max(abs(a["cost"]), 1e-300) for a, b in list(zip(t0, t1))[:min(n, 4)]), default=0.0)
      dc = abs(s0["final_cost"] - s1["final_cost"]) / max(s0["final_cost"], 1e-300)
      dx = float(np.abs(x0 - x1).max())
      # after the first few iterations the two implementations may take different accept / reject decisions on
      # ill-conditioned windows; the first iterations and the counts of the initial evaluation must agree
      if abs(s0["initial_cost"] - s1["initial_cost"]) > 1e-11 * max(s0["initial_cost"], 1e-300) or dtr > 1e-6 or \
         s0["num_free_parameters"] != s1["num_free_parameters"] or s0["num_residual_blocks"] != s1["num_residual_blocks"]:
          ok = False
      if not ok or dc > 1e-3:
          bad += not ok
          if verbose: print("case %d %s kf=%d free=%d L=%d M=%d: init %.3e/%.3e final %.6e/%.6e steps %d+%d / %d+%d term %d/%d dtrace %.1e dx %.1e %s" % (
              case, "FAIL" if not ok else "diverged-late", nkf, nfree, w["num_lines"], m, s0["initial_cost"], s1["initial_cost"],
              s0["final_cost"], s1["final_cost"], s0["num_successful_steps"], s0["num_unsuccessful_steps"],
              s1["num_successful_steps"], s1["num_unsuccessful_steps"], s0["termination_type"], s1["termination_type"], dtr, dx, opt))
      worst["cost"] = max(worst["cost"], dc); worst["x"] = max(worst["x"], dx); worst["trace"] = max(worst["trace"], dtr)
  return bad, worst


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    bad, worst = run(cases, int(os.environ.get("FUZZ_SEED", "2026")), oversize=bool(int(os.environ.get("FUZZ_OVERSIZE", "0"))))
    print("cases %d, hard failures %d, worst rel final-cost diff %.2e, worst |dx| %.2e, worst early-trace diff %.2e" % (
        cases, bad, worst["cost"], worst["x"], worst["trace"]))
    sys.exit(1 if bad else 0)
