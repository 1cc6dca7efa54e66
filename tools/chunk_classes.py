"""Chunk durations of one elimination (or back-substitution) launch by dispatch class and by position inside the window (wall-stamp build,
see tools/chunk_timeline.py): are the chunks of a class equal in TIME?   python tools/chunk_classes.py [windows] [sweep] [elimination|backsub]"""
import os, sys, ctypes, json
os.environ["SLSLAM_DEBUG_ABLATE"] = "8192"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slslam_amd import capi, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
it = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kernel = sys.argv[3] if len(sys.argv) > 3 else "elimination"
ws = [synth.make_window(i, num_lines=2000) for i in range(min(nwin, 64))]
b = capi.LBABatch()
for i in range(nwin): b.add(ws[i % len(ws)])
b.finalize(use_graph=0, max_num_iterations=it)
b.solve(); b.download(); b.reset(); b.solve(); b.download()
size = ctypes.c_longlong(0)
capi.lib().slslam_debug_read_cycles(b._h, None, 0, ctypes.byref(size))
raw = np.zeros(size.value, dtype=np.uint64)
capi.lib().slslam_debug_read_cycles(b._h, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), size.value, ctypes.byref(size))
code = b.window_chunks(0)
cpw = abs(code) % 1000
nchunk = cpw * nwin
t = raw[:32 * nchunk].reshape(nchunk, 32)
s0, s1 = (30, 31) if kernel == "elimination" else (28, 29)
start, end = t[:, s0].astype(np.float64) * 0.01, t[:, s1].astype(np.float64) * 0.01
t0 = start[end > start].min()
start -= t0; end -= t0
dur = end - start
print("cut code", code, "chunks", nchunk, "makespan %.1f us, perfectly packed %.1f us" % (end.max(), dur.sum() / 2048))
# stamps are indexed by the chunk id: window-major, chunk c of window w = w * cpw + c (plan_layout); dispatch class of chunk c = c * rounds / cpw
rounds = abs(code) // 1000 if code < 0 else 1
D, S, E = dur.reshape(nwin, cpw), start.reshape(nwin, cpw), end.reshape(nwin, cpw)
for c in range(cpw):
    d, s_, e = D[:, c], S[:, c], E[:, c]
    print("chunk %d of a window (class %d): duration mean %.1f std %.1f min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | start p1 %.1f p50 %.1f p99 %.1f | end p50 %.1f p99 %.1f max %.1f" % (
        c, c * rounds // cpw, d.mean(), d.std(), d.min(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max(),
        np.percentile(s_, 1), np.percentile(s_, 50), np.percentile(s_, 99), np.percentile(e, 50), np.percentile(e, 99), e.max()))

# is a long chunk's duration a property of its WINDOW (the batch repeats 64 shapes) or of where it ran?
nshape = len(ws)
if nwin >= 4 * nshape:
    d0 = D[:, 0].reshape(-1, nshape)                    # [repeat, shape]
    print("chunk 0 by window shape: mean over shapes of (std across the repeats of a shape) %.1f us; std across shapes of (mean of a shape) %.1f us; overall std %.1f" % (
        d0.std(axis=0).mean(), d0.mean(axis=0).std(), D[:, 0].std()))
    steps = np.array([b.summary(i)["num_unsuccessful_steps"] for i in range(nshape)])
    order = np.argsort(d0.mean(axis=0))
    print("shapes sorted by mean duration of chunk 0 (us | rejected steps of the solve | observations):")
    print("  " + "  ".join("%.0f|%d|%d" % (d0.mean(axis=0)[k], steps[k], len(ws[k]["camera_index"])) for k in order[::4]))
# where the waves ran (wall build, word start - 4): XCC_ID << 32 | HW_ID (wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
hwid = t[:, s0 - 4]
if hwid.any():
    xcc, hw = (hwid >> np.uint64(32)).astype(np.int64) & 15, (hwid & np.uint64(0xffffffff)).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    long_ = np.zeros(nchunk, bool); long_.reshape(nwin, cpw)[:, :max(1, cpw // rounds)] = True
    for name, key in (("XCD", xcc), ("SE", se), ("SIMD", simd), ("CU in its SE/SH", cu * 2 + sh)):
        vals = sorted(set(key[long_].tolist()))
        print("long chunks by %s: " % name + "  ".join("%d: %.0f (%d)" % (v, dur[long_ & (key == v)].mean(), (long_ & (key == v)).sum()) for v in vals))
    cuid = (xcc * 8 + se) * 32 + cu * 2 + sh
    m = np.array([dur[long_ & (cuid == v)].mean() for v in sorted(set(cuid[long_].tolist()))])
    print("long chunks by CU (%d CUs seen): mean of the per-CU means %.1f, std %.1f, min %.1f, max %.1f; std inside a CU (mean over CUs) %.1f" % (
        len(m), m.mean(), m.std(), m.min(), m.max(), np.mean([dur[long_ & (cuid == v)].std() for v in sorted(set(cuid[long_].tolist()))])))
if hwid.any():
    # the two long chunks that share a SIMD from t = 0: equal partners, or one ahead of the other (issue arbitration: oldest wave first)?
    simd_id = cuid * 4 + simd
    wave = hw & 15
    lo, hi, w_lo = [], [], []
    for v in sorted(set(simd_id[long_].tolist())):
        sel = np.where(long_ & (simd_id == v))[0]
        if len(sel) == 2:
            a, c = (sel[0], sel[1]) if dur[sel[0]] <= dur[sel[1]] else (sel[1], sel[0])
            lo.append(dur[a]); hi.append(dur[c]); w_lo.append(int(wave[a] < wave[c]))
    lo, hi = np.array(lo), np.array(hi)
    print("SIMDs with two long chunks: %d; faster of the pair mean %.1f std %.1f, slower mean %.1f std %.1f; the faster one has the lower wave slot id in %.0f %%; sum of the pair mean %.1f std %.1f" % (
        len(lo), lo.mean(), lo.std(), hi.mean(), hi.std(), 100.0 * np.mean(w_lo), (lo + hi).mean(), (lo + hi).std()))
if hwid.any():
    # the gap between two chunks on the same wave slot: end of one workgroup -> first instruction of the next one dispatched there
    slot = simd_id * 16 + wave
    gaps, per_slot = [], []
    for v in sorted(set(slot.tolist())):
        sel = np.where(slot == v)[0]
        sel = sel[np.argsort(start[sel])]
        per_slot.append(len(sel))
        for a, c in zip(sel[:-1], sel[1:]):
            gaps.append(start[c] - end[a])
    gaps = np.array(gaps)
    print("wave slots seen %d (chunks per slot min %d max %d); gap between consecutive chunks of a slot: mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f, max %.2f (n = %d)" % (
        len(per_slot), min(per_slot), max(per_slot), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90), gaps.max(), len(gaps)))
    last_end = np.array([end[slot == v].max() for v in sorted(set(slot.tolist()))])
    print("last chunk of a slot ends at: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f -> idle at the end of the launch: %.1f %% of the slot-time" % (
        last_end.mean(), np.percentile(last_end, 10), np.percentile(last_end, 50), np.percentile(last_end, 90), last_end.max(), 100.0 * (1.0 - last_end.mean() / last_end.max())))
b.close()
