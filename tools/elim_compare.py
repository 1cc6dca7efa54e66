"""A/B of the elimination sweeps on the bench batch (same windows, one process): per-kernel hipEvent times and it/s.
usage: python tools/elim_compare.py [--windows 1024] [--lines 2000] [--modes 1,2,3] [--steps 5]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1024)
ap.add_argument("--lines", type=int, default=2000)
ap.add_argument("--modes", default="1,3,2")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--chunks", type=int, default=0)
ap.add_argument("--ablate", default="0", help="SLSLAM_DEBUG_ABLATE values to run each mode with (timing experiments; results wrong when != 0)")
args = ap.parse_args()
wins = [synth.make_window(i, num_lines=args.lines) for i in range(args.windows)]
ref = None
for mode, abl in [(int(m), a) for m in args.modes.split(",") for a in args.ablate.split(",")]:
    os.environ["SLSLAM_DEBUG_ABLATE"] = abl
    bt = capi.LBABatch()
    for w in wins:
        bt.add(w)
    bt.finalize(use_graph=0, lba_elimination=mode, chunks_per_window=args.chunks)
    bt.set_profiling(True)
    for _ in range(2):
        bt.reset(); bt.solve()
    torch.cuda.synchronize()
    bt.iterations(clear=True)
    bt.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bt.reset(); bt.solve()
    its = bt.iterations()
    dt = time.perf_counter() - t0
    bt.download()
    kt = bt.kernel_times()
    x0 = bt.parameters(0)
    s0 = bt.summary(0)
    if ref is None:
        ref = (x0, s0)
    out = {"mode": mode, "ablate": abl, "it_per_s": its / dt, "ms_per_step": 1e3 * dt / args.steps, "lm_iterations": its,
           "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in kt.items() if v[1] > 0},
           "win0": {"steps": (s0["num_successful_steps"], s0["num_unsuccessful_steps"]), "final_cost": s0["final_cost"],
                    "max_param_diff_vs_first_mode": float(np.abs(x0 - ref[0]).max())}}
    if int(abl) & 256:
        import ctypes
        ph = np.zeros(16)
        capi.lib().slslam_debug_phase_cycles(bt._h, ph.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        names = ["rows", "cam_block", "gram_line", "seg_sum", "chol", "Z_F_panel_b", "prefetch_issue", "barrier1", "mfma", "barrier2"]
        out["phase_share"] = {n: round(float(ph[i] / ph[:10].sum()), 4) for i, n in enumerate(names)}
        out["phase_cycles_per_tile_wave"] = {n: round(float(ph[i]) / (sum(len(w["line_index"]) for w in wins) / 62.0), 1) for i, n in enumerate(names)}
    print(json.dumps(out), flush=True)
    bt.close()
