"""Where the window build (csrc/lba_device_build.h: k_build_lines, k_build_rows, k_build_order) spends its time: constant-clock (100 MHz) stamps of its
phases for one window alone (a stamp pair that spans two kernels - "bitonic sort" ends in the next kernel - includes the launch gap)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402

NAMES = ["pass 1: counts, masks (LDS atomics)", "line records + sort keys", "bitonic sort of the lines", "rows: best fit (one wave)",
         "row order + tile ranges", "line order (chains), flags", "line pointers, positions", "observation keys scattered",
         "per-line key sort, ob_orig / ob_cam", "line-level arrays"]
for lines in (2000, 500):
    w = synth.make_window(0, num_lines=lines)
    for g in (1, 0):
        clk = np.zeros(16, dtype=np.uint64)
        capi.debug_device_pack(w, grouping=g)
        st, D = capi.debug_device_pack(w, grouping=g, clocks=clk)
        t = clk[:10].astype(np.int64)
        d = np.diff(t)
        tot = int(t[9] - t[0])
        print("%d lines, grouping %d: %.1f us from the first stamp to the last (three kernels, launch gaps included), %d tiles" % (lines, g, tot / 100.0, D["ntiles"]))
        for n, c in zip(NAMES[:9], d):
            print("   %-44s %9.1f us  %5.1f %%" % (n, c / 100.0, 100.0 * c / max(1, tot)))
        tt = clk[10:15].astype(np.int64)
        if tt[4] > tt[0] > 0:
            print("   k_build_tiles: prologue (tile starts, scans) %.1f us, inputs into LDS %.1f us, tile loop %.1f us, descriptors out %.1f us"
                  % tuple((b - a) / 100.0 for a, b in zip(tt[:-1], tt[1:])))
