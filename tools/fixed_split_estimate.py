"""What a split of every line's observations into a free-camera run and a fixed-camera run could buy (timing experiment).
(a) the bench batch as it is; (b) the same windows with the fixed-camera observations dropped: the sweeps then see what the
free-camera tiles of a split layout would see (denser camera-pair work per tile, no idle camera-side lanes).  The
fixed-camera tiles would add about (their tile count) x (the line-side share of a tile) on top of (b).
usage: python tools/fixed_split_estimate.py [--windows 1024]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1024)
ap.add_argument("--lines", type=int, default=2000)
ap.add_argument("--steps", type=int, default=4)
args = ap.parse_args()


def drop_fixed(w):
    cam = np.asarray(w["camera_index"]); fx = np.asarray(w["fixed_index"]).reshape(-1, 2)
    camfixed = np.zeros(w["num_cameras"], bool)
    np.logical_or.at(camfixed, cam, fx[:, 0] != 0)
    keep = ~camfixed[cam]
    return dict(w, camera_index=cam[keep], line_index=np.asarray(w["line_index"])[keep], fixed_index=fx[keep].reshape(-1),
                observations=np.asarray(w["observations"])[keep]), int(keep.sum()), int((~keep).sum())


wins = [synth.make_window(i, num_lines=args.lines) for i in range(args.windows)]
variants = {"as_is": wins}
dropped = [drop_fixed(w) for w in wins]
variants["free_camera_observations_only"] = [d[0] for d in dropped]
print(json.dumps({"free_obs_per_window": float(np.mean([d[1] for d in dropped])), "fixed_obs_per_window": float(np.mean([d[2] for d in dropped]))}))
for name, ws in variants.items():
    bt = capi.LBABatch()
    for w in ws:
        bt.add(w)
    bt.finalize(use_graph=0)
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
    print(json.dumps({"variant": name, "lm_iterations_per_step": its / args.steps, "ms_per_step": 1e3 * dt / args.steps,
                      "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in kt.items() if v[1] > 0},
                      "launches": {k: v[1] for k, v in kt.items() if v[1] > 0}}), flush=True)
    bt.close()
