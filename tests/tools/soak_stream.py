"""Differential soak of the STREAMED path (slslam_lba_stream_*: refillable batches, pinned staging, tile contexts built on the device): batches of
random window shapes - other free-camera counts, track lengths, line counts, constant lines, scrambled observation order from one batch to the
next, so that refills reuse arrays that held something else and some batches do not fit and are rebuilt - with slslam_solver_options.reproducible = 1,
under which a window's bytes are a function of the window alone: every streamed window must equal the same window solved alone, bit for bit, and
a sample of them is held against the CPU oracle (checker).   python tests/tools/soak_stream.py [batches] [windows per batch] [seed] [lba_elimination 0 | 4] [arrays: 0 pageable | 1 page-locked | 2 page-locked, indices narrowed by the caller]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from slslam_amd import capi, synth
from oracle import pyoracle

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
ELIM = int(sys.argv[4]) if len(sys.argv) > 4 else 0
max_free = 10 if ELIM == 4 else 20
ARR = int(sys.argv[5]) if len(sys.argv) > 5 else 0


def random_window():
    free = int(rng.integers(2, max_free + 1))
    kf = free + int(rng.integers(0, free + 3))
    lines = int(rng.integers(40, 420))
    w = synth.make_window(int(rng.integers(1, 10 ** 6)), num_lines=lines, num_kf=kf, num_free=free, mean_track=float(rng.uniform(3.0, max(3.5, 0.8 * kf))))
    m = len(w["camera_index"])
    if rng.random() < 0.4:
        perm = rng.permutation(m)
        for k in ("camera_index", "line_index", "observations"):
            w[k] = np.asarray(w[k])[perm]
        w["fixed_index"] = np.asarray(w["fixed_index"]).reshape(-1, 2)[perm].reshape(-1)
    if rng.random() < 0.25:
        const = rng.random(w["num_lines"]) < 0.2
        fi = np.asarray(w["fixed_index"]).reshape(-1, 2).copy(); fi[:, 1] = const[np.asarray(w["line_index"])]; w["fixed_index"] = fi.reshape(-1)
    return w


t0 = time.time()
sets = [[random_window() for _ in range(per)] for _ in range(nb)]
st = capi.LBAStream(depth=3, host_threads=4, reproducible=1, lba_elimination=ELIM)
wsets = [capi.WindowSet(s, pinned=ARR >= 1, packed=ARR == 2) for s in sets]
tickets, summaries = [], {}
for k in range(nb):
    if k >= 3:
        summaries[k - 3] = st.collect(tickets[k - 3])
    tickets.append(st.submit(wsets[k]))
for k in range(max(0, nb - 3), nb):
    summaries[k] = st.collect(tickets[k])
stats = st.stats()
bstats = st.build_stats()
st.close()
diff = bad = checked = 0
for k in range(nb):
    for i, w in enumerate(sets[k]):
        x, s, _ = capi.lba_solve(w, reproducible=1, lba_elimination=ELIM)
        if not (np.array_equal(wsets[k].parameters(i), x) and summaries[k][i] == s):
            diff += 1
            print("DIFF batch %d window %d (cams %d lines %d obs %d): max |dx| %.3e" % (k, i, w["num_cameras"], w["num_lines"], len(w["camera_index"]), np.abs(wsets[k].parameters(i) - x).max()))
        if (k * per + i) % 7 == 0:
            xo, so, _ = pyoracle.lba_solve(w, linear_solver=1)
            checked += 1
            hard = [q for q in ("num_successful_steps", "num_unsuccessful_steps", "termination_type") if so[q] != s[q]]
            rel = abs(so["final_cost"] - s["final_cost"]) / max(abs(so["final_cost"]), 1e-300)
            if hard or rel > 1e-5 or np.abs(xo - x).max() > 1e-4:
                bad += 1
                print("ORACLE batch %d window %d: %s, final cost rel. diff %.2e, max |dx| %.2e" % (k, i, hard, rel, np.abs(xo - x).max()))
print("streamed %d batches x %d random windows (lba_elimination %d, reproducible): %d differ from their solo solve; %d of %d sampled windows off the oracle; builds %d, refills %d; %.1f s" % (
    nb, per, ELIM, diff, bad, checked, stats["builds"], stats["refills"], time.time() - t0))
print("   arrays: %s; %s" % (("pageable", "page-locked", "page-locked, indices narrowed by the caller")[ARR], bstats))
sys.exit(1 if (diff or bad) else 0)
