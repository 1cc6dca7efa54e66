"""Soak of the device build (csrc/lba_device_build.h) against the host packer: random window shapes - keyframe counts, free cameras, track
lengths, scrambled observation order, constant lines and cameras, unobserved lines - both packings, every emitted array compared byte for byte.
    python tests/tools/soak_device_build.py [n_windows] [seed]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from slslam_amd import capi, synth  # noqa: E402
from test_host_side import _pack  # noqa: E402

FIELDS = ("line_order", "line_ptr", "ob_orig", "ob_cam", "cam_cf", "tiles", "items", "lane_map", "desc")


def host_math():
    out = os.path.join(ROOT, "tests", "_build", "libhost_math.so")
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "host_math", "host_math.cpp"),
                               os.path.join(ROOT, "slslam_amd", "csrc", "lba_pack.cpp")])
    return C.CDLL(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    hm = host_math()
    done = flagged = 0
    for k in range(n):
        kf = int(rng.integers(3, 65))
        free = int(rng.integers(2, min(kf, 22) + 1)) if kf >= 2 else 2
        lines = int(rng.choice([1, 3, 17, 60, 150, 400, 900, 2000, 3000]))
        track = float(rng.choice([1.2, 2.0, 3.5, 6.0, 12.0, 25.0, 50.0]))
        w = synth.make_window(int(rng.integers(1 << 30)), num_lines=lines, num_kf=kf, num_free=free, mean_track=track)
        M = len(w["camera_index"])
        if rng.random() < 0.5 and M > 1:
            perm = rng.permutation(M)
            w["camera_index"] = np.asarray(w["camera_index"])[perm]; w["line_index"] = np.asarray(w["line_index"])[perm]
            w["observations"] = np.asarray(w["observations"]).reshape(-1, 8)[perm].reshape(-1)
            w["fixed_index"] = np.asarray(w["fixed_index"]).reshape(-1, 2)[perm].reshape(-1)
        fx = np.asarray(w["fixed_index"]).reshape(-1, 2).copy()
        if rng.random() < 0.4:
            const = rng.random(w["num_lines"]) < rng.choice([0.05, 0.3, 0.9])
            fx[:, 1] = const[np.asarray(w["line_index"])]
        if rng.random() < 0.3 and M > 0:
            fx[np.asarray(w["camera_index"]) == int(rng.integers(w["num_cameras"])), 0] = 1
        w["fixed_index"] = fx.reshape(-1)
        if rng.random() < 0.2:
            extra = int(rng.integers(1, 12))
            w = dict(w, num_lines=w["num_lines"] + extra, parameters=np.concatenate([w["parameters"], np.tile([0.1, 0.2, 0.3, 0.4], extra)]))
        for g in (0, 1):
            rc, P = _pack(hm, w, grouping=g)
            assert rc == 0
            st, D = capi.debug_device_pack(w, grouping=g)
            big = P["Cf"] > 20
            if st != 0:
                assert st == 2 and big, (k, g, st, P["Cf"])
                flagged += 1
                continue
            assert not big
            for q in ("Cf", "ntiles", "nitems", "nfree", "nkept"):
                assert D[q] == P[q], (k, g, q, D[q], P[q], kf, free, lines, track)
            for q in FIELDS:
                a, b = np.asarray(D[q]), np.asarray(P[q])
                assert a.shape == b.shape and np.array_equal(a, b), (k, g, q, kf, free, lines, track)
            done += 1
    print("device build == host packer on %d packings of %d random windows (%d flagged for the host path, as the host packer's shape test says)" % (done, n, flagged))


if __name__ == "__main__":
    main()
