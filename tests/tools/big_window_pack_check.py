"""Device build against the host packer on windows near the limits of the device path: thousands of lines (more than 256 tiles: several passes of
k_build_tiles' thread <-> tile loop), long tracks, many observations (the staged / unstaged forms of k_build_tiles by what fits the LDS)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from slslam_amd import capi, synth  # noqa: E402
import test_host_side as T  # noqa: E402
import test_gpu_device_build as G  # noqa: E402

hm = C.CDLL(os.path.join(ROOT, "tests", "_build", "libhost_math.so"))
n = 0
for seed, kw in ((1, dict(num_lines=3000)), (2, dict(num_lines=4000, mean_track=6.0)), (3, dict(num_lines=4500, mean_track=4.0)),
                 (4, dict(num_lines=2500, num_kf=40, num_free=20, mean_track=30.0)), (5, dict(num_lines=1200, num_kf=64, num_free=10, mean_track=50.0)),
                 (6, dict(num_lines=3500, num_kf=30, num_free=15, mean_track=12.0))):
    w = synth.make_window(seed, **kw)
    for g in (0, 1):
        D = G._compare(capi, hm, w, g, "big%d" % seed)
        n += 1
        print("seed %d grouping %d: %d lines, %d observations, %d tiles: equal" % (seed, g, w["num_lines"], len(w["camera_index"]), D["ntiles"]))
print("device build == host packer on %d large packings" % n)
