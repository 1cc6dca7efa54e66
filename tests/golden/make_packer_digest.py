"""Generates tests/golden/packer_digest.json: crc32 of everything pack_window emits (line order, line_ptr, ob_orig, ob_cam, cam_cf, tiles,
items, lane_map, descriptors) for a fixed family of synthetic windows, both packings.  Pins the packed LAYOUT: a window's solved bytes are
a function of it (summation order), and tests/golden/bench_digest.json pins those on the GPU only.  Made with the packer of commit
6674089 whose output was compared, field by field, with the packer before its host-side diet on 166 windows (round 5).
    python tests/golden/make_packer_digest.py"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from slslam_amd import synth  # noqa: E402


def family():
    ws = {"bench_0_2000": synth.make_window(0, num_lines=2000), "bench_3_500": synth.make_window(3, num_lines=500),
          "wide_10": synth.make_window(10, num_lines=60, num_kf=24, num_free=10, mean_track=40.0),
          "many_free_11": synth.make_window(11, num_lines=300, num_kf=40, num_free=20, mean_track=12.0),
          "motion_only_13": synth.make_motion_only(13, num_lines=100)}
    rng = np.random.default_rng(5)
    w = synth.make_window(21, num_lines=240, mean_track=2.0)            # track length 2: camera ranges with holes, scrambled order
    perm = rng.permutation(len(w["camera_index"]))
    for k in ("camera_index", "line_index", "observations"):
        w[k] = w[k][perm]
    w["fixed_index"] = w["fixed_index"].reshape(-1, 2)[perm].reshape(-1)
    const = rng.random(w["num_lines"]) < 0.3
    fi = w["fixed_index"].reshape(-1, 2).copy(); fi[:, 1] = const[w["line_index"]]; w["fixed_index"] = fi.reshape(-1)
    ws["scrambled_21"] = w
    return ws


def digest(P):
    crc = 0
    for k in ("line_order", "line_ptr", "ob_orig", "ob_cam", "cam_cf", "tiles", "items", "lane_map", "desc"):
        crc = zlib.crc32(np.ascontiguousarray(P[k]).tobytes(), crc)
    return "%08x" % crc


if __name__ == "__main__":
    import ctypes as C
    from test_host_side import _pack      # noqa: E402
    # (tests/_build/libhost_math.so is built by the `host_math` fixture of tests/conftest.py: run the CPU suite once first)
    hm = C.CDLL(os.path.join(os.path.dirname(HERE), "_build", "libhost_math.so"))
    out = {}
    for name, w in family().items():
        for g in (0, 1):
            rc, P = _pack(hm, w, grouping=g)
            assert rc == 0
            out["%s/grouping%d" % (name, g)] = {"crc32": digest(P), "tiles": P["ntiles"], "items": P["nitems"]}
    json.dump(out, open(os.path.join(HERE, "packer_digest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
