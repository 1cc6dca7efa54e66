#!/usr/bin/env python3
"""Tile statistics of the bench window (host only, no GPU): python tools/pack_stats.py [seed] [lines]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import synth  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = synth.make_window(seed, num_lines=lines)
d = tempfile.mkdtemp()
np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"])], dtype=np.int32).tofile(d + "/hdr.bin")
np.asarray(w["camera_index"], dtype=np.int32).tofile(d + "/cam.bin")
np.asarray(w["line_index"], dtype=np.int32).tofile(d + "/line.bin")
np.asarray(w["fixed_index"], dtype=np.int32).tofile(d + "/fixed.bin")
np.asarray(w["observations"], dtype=np.float64).tofile(d + "/obs.bin")
np.asarray(w["parameters"], dtype=np.float64).tofile(d + "/par.bin")
csrc = os.path.join(ROOT, "slslam_amd", "csrc")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + csrc, os.path.join(ROOT, "tools", "pack_stats.cpp"),
                       os.path.join(csrc, "lba_pack.cpp"), "-o", d + "/pack_stats"])
subprocess.check_call([d + "/pack_stats"], cwd=d)
