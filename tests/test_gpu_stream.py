"""GPU tests of the streamed form of BASELINE config 4: a batch whose windows are REPLACED (slslam_lba_batch_refill) and the stream
object on top of it (slslam_lba_stream_*) - what a caller that hands over five host arrays per window (reference
src/slam.cpp:899-921) drives instead of building a batch per set of windows.  The bar is bytes: a refilled batch returns what a
fresh batch of the same windows returns."""
import os

import numpy as np
import pytest

from slslam_amd import synth

pytestmark = pytest.mark.gpu


def _solve_fresh(hip, ws, **opt):
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize(**opt)
    b.solve(); b.download()
    out = [(b.parameters(i).copy(), b.summary(i), b.trace(i)) for i in range(len(ws))]
    cuts = [b.window_chunks(i) for i in range(len(ws))]
    elim = b.elimination()
    b.close()
    return out, cuts, elim


@pytest.mark.parametrize("elim", [1, 4])
def test_refilled_batch_equals_fresh_batch(hip, oracle, elim):
    """Three sets of windows of differing sizes through ONE batch (finalize once, refill twice, back to the first set) against three
    fresh batches: parameters, summaries and iteration traces identical to the byte, chunk cuts equal, the captured graph reused; the
    first refilled set also against the oracle.  A set that does not fit is refused and leaves the batch solvable."""
    sets = [[synth.make_window(100 * k + i, num_lines=n) for i, n in enumerate((300, 420, 380, 350, 400, 330))] for k in range(3)]
    b = hip.LBABatch()
    for w in sets[0]:
        b.add(w)
    b.finalize(lba_elimination=elim, refill_headroom_percent=25, host_threads=4)
    assert b.elimination() == elim
    order = [0, 1, 2, 0]
    for step, k in enumerate(order):
        if step > 0:
            b.refill(sets[k])
        b.solve(); b.download()
        fresh, cuts, e = _solve_fresh(hip, sets[k], lba_elimination=elim)
        assert e == elim
        for i, w in enumerate(sets[k]):
            assert b.window_chunks(i) == cuts[i]
            assert np.array_equal(b.parameters(i), fresh[i][0]), "set %d window %d" % (k, i)
            assert b.summary(i) == fresh[i][1] and b.trace(i) == fresh[i][2]
        b.reset(); b.solve(); b.download()                  # a refilled batch is reset from ITS windows' initial values
        for i in range(len(sets[k])):
            assert np.array_equal(b.parameters(i), fresh[i][0])
    for i in (0, 3):
        xo, so, _ = oracle.lba_solve(sets[0][i], linear_solver=1)
        assert so["num_successful_steps"] == b.summary(i)["num_successful_steps"]
        assert abs(so["final_cost"] - b.summary(i)["final_cost"]) <= 1e-7 * so["final_cost"] and np.abs(xo - b.parameters(i)).max() < 1e-5
    keep = [b.parameters(i).copy() for i in range(6)]
    too_big = [synth.make_window(900 + i, num_lines=900) for i in range(6)]
    with pytest.raises(hip.SlslamError) as e:
        b.refill(too_big)
    assert e.value.status == 4
    with pytest.raises(hip.SlslamError):
        b.refill(sets[1][:5])                               # another number of windows
    b.reset(); b.solve(); b.download()
    for i in range(6):
        assert np.array_equal(b.parameters(i), keep[i])     # the refused refills left the batch as it was
    b.close()
    plain = hip.LBABatch()
    for w in sets[0]:
        plain.add(w)
    plain.finalize()
    with pytest.raises(hip.SlslamError) as e:
        plain.refill(sets[1])                               # not finalized for refills
    assert e.value.status == 4
    plain.close()


def test_refill_edge_shapes(hip, oracle):
    """Windows whose shape changes between refills: other free-camera counts (another order of the reduced system: its map table must be
    there), constant lines, scrambled observation order, fewer lines."""
    first = [synth.make_window(10 + i, num_lines=160, num_kf=20, num_free=10) for i in range(4)]
    b = hip.LBABatch()
    for w in first:
        b.add(w)
    b.finalize(lba_elimination=1, refill_headroom_percent=60, chunks_per_window=2)
    other = [synth.make_window(20, num_lines=120, num_kf=12, num_free=6), synth.make_window(21, num_lines=200, num_kf=20, num_free=10),
             synth.make_window(22, num_lines=90, num_kf=8, num_free=3), synth.make_window(23, num_lines=150, num_kf=16, num_free=8)]
    rng = np.random.default_rng(7)
    w = dict(other[1]); perm = rng.permutation(len(w["camera_index"]))
    w["camera_index"] = np.asarray(w["camera_index"])[perm]; w["line_index"] = np.asarray(w["line_index"])[perm]
    w["observations"] = np.asarray(w["observations"]).reshape(-1, 8)[perm].reshape(-1)
    fx = np.asarray(w["fixed_index"]).reshape(-1, 2)[perm].copy(); fx[np.isin(w["line_index"], (3, 9, 40)), 1] = 1
    w["fixed_index"] = fx.reshape(-1)
    other[1] = w
    b.refill(other)
    b.solve(); b.download()
    fresh, cuts, _ = _solve_fresh(hip, other, lba_elimination=1, chunks_per_window=2)
    for i, w in enumerate(other):
        assert np.array_equal(b.parameters(i), fresh[i][0]) and b.summary(i) == fresh[i][1]
        xo, so, _ = oracle.lba_solve(w, linear_solver=1)
        assert so["num_successful_steps"] == b.summary(i)["num_successful_steps"] and np.abs(xo - b.parameters(i)).max() < 1e-5
    b.close()


def test_stream_of_windows(hip, oracle):
    """slslam_lba_stream_*: eight batches of 12 windows through a depth-3 stream (submit / collect in ticket order, the solved parameters
    written in place), every window equal to the byte to the same window in a fresh batch of its set; submits after the first `depth`
    are refills; collecting out of turn and leaving a slot uncollected are refused."""
    nb, per = 8, 12
    sets = [[synth.make_window(5000 + 100 * k + i, num_lines=260 + 10 * (i % 4)) for i in range(per)] for k in range(nb)]
    st = hip.LBAStream(depth=3, host_threads=4)
    wsets = [hip.WindowSet(s) for s in sets]
    tickets, results = [], {}
    for k in range(nb):
        if k >= 3:
            results[tickets[k - 3]] = st.collect(tickets[k - 3])
        tickets.append(st.submit(wsets[k]))
    with pytest.raises(hip.SlslamError):
        st.collect(tickets[0])                              # collected already
    for k in range(nb - 3, nb):
        results[tickets[k]] = st.collect(tickets[k])
    stats = st.stats()
    assert stats["builds"] == 3 and stats["refills"] == nb - 3 and stats["windows"] == nb * per
    for k in range(nb):
        fresh, _, _ = _solve_fresh(hip, sets[k])
        for i in range(per):
            assert np.array_equal(wsets[k].parameters(i), fresh[i][0]), "batch %d window %d" % (k, i)
            s = results[tickets[k]][i]
            assert s == fresh[i][1]
    xo, so, _ = oracle.lba_solve(sets[5][7], linear_solver=1)
    assert np.abs(xo - wsets[5].parameters(7)).max() < 1e-5
    # a slot that has not been collected cannot be submitted to again
    t = [st.submit(wsets[k]) for k in range(3)]
    with pytest.raises(hip.SlslamError) as e:
        st.submit(wsets[3])
    assert e.value.status == 5
    for x in t:
        st.collect(x)
    st.close()


def test_reproducible_option_makes_results_independent_of_the_batch(hip, oracle):
    """slslam_solver_options.reproducible: sweep and chunk cut are functions of the window alone, so a window
    returns the same bytes alone, in a small batch, in a large batch and through a stream - with no caller bookkeeping (without the
    option the automatic cut depends on the batch: asserted too, as the reason the option exists)."""
    ws = [synth.make_window(300 + i, num_lines=n) for i, n in enumerate((2000, 700, 1500, 300, 1100, 2000, 900, 450))]
    many = ws + [synth.make_window(400 + i, num_lines=600) for i in range(56)]
    for elim in (0, 4):
        big, cuts_big, e_big = _solve_fresh(hip, many, reproducible=1, lba_elimination=elim)
        small, cuts_small, e_small = _solve_fresh(hip, ws[:3], reproducible=1, lba_elimination=elim)
        assert e_big == e_small == (1 if elim == 0 else 4)
        assert cuts_big[:3] == cuts_small
        assert cuts_big[0] == -3006                         # a 2000-line window: six chunks graded for three rounds, the headline's cut
        for i in range(3):
            assert np.array_equal(big[i][0], small[i][0]) and big[i][1] == small[i][1]
        for i in (0, 3, 6):
            x, s, t = hip.lba_solve(ws[i], reproducible=1, lba_elimination=elim)
            assert np.array_equal(x, big[i][0]) and s == big[i][1]
    auto_big, cuts_a, _ = _solve_fresh(hip, many)
    auto_one, cuts_b, _ = _solve_fresh(hip, ws[:1])
    assert cuts_a[0] != cuts_b[0]                            # the automatic cut follows the batch
    xo, so, _ = oracle.lba_solve(ws[1], linear_solver=1)
    assert np.abs(xo - big[1][0]).max() < 1e-5
    # a requested matrix-core sweep the batch cannot take is refused instead of replaced
    wide = synth.make_window(41, num_lines=60, num_kf=30, num_free=14)
    with pytest.raises(hip.SlslamError) as e:
        hip.lba_solve(wide, reproducible=1, lba_elimination=4)
    assert e.value.status == 4
    hip.lba_solve(wide, reproducible=1)


def test_stream_from_a_cxx_host(hip, tmp_path):
    """tests/host_cxx/stream_demo.cpp: the stream API driven from C++ through the C ABI alone (no ctypes, no torch) - six batches of eight windows
    of three shapes through a depth-2 stream on three host threads, the later batches refills; with reproducible = 1 every solved parameter
    vector equals, to the byte, the same window solved alone through slslam_lba_solve, and the step count the demo adds up (the counter of
    reference src/slam.cpp:949-950) equals the sum of the solo summaries."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(ROOT, "slslam_amd", "_lib")
    demo = os.path.join(ROOT, "tests", "_build", "stream_demo")
    os.makedirs(os.path.dirname(demo), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", demo,
                           os.path.join(ROOT, "tests", "host_cxx", "stream_demo.cpp"), "-L", libdir, "-lslslam_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    per, nb = 8, 6
    ws = [synth.make_window(900 + i, num_lines=(150, 220, 300)[i % 3], num_kf=(12, 20, 16)[i % 3], num_free=(6, 10, 8)[i % 3]) for i in range(per * 3)]
    with open(tmp_path / "wins.bin", "wb") as f:
        np.array([len(ws)], dtype=np.int32).tofile(f)
        for w in ws:
            np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"])], dtype=np.int32).tofile(f)
            np.asarray(w["camera_index"], dtype=np.int32).tofile(f)
            np.asarray(w["line_index"], dtype=np.int32).tofile(f)
            np.asarray(w["fixed_index"], dtype=np.int32).tofile(f)
            np.asarray(w["observations"], dtype=np.float64).tofile(f)
            np.asarray(w["parameters"], dtype=np.float64).tofile(f)
    p = subprocess.run([demo, str(tmp_path / "wins.bin"), str(tmp_path / "out.bin"), str(per), str(nb), "2", "3"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "refilled" in p.stdout
    out = np.fromfile(tmp_path / "out.bin")
    steps, off = out[0], 1
    solo = {}
    want_steps = 0
    for k in range(nb):
        for j in range(per):
            i = (k * per + j) % len(ws)
            if i not in solo:
                solo[i] = hip.lba_solve(ws[i], reproducible=1)
            x, s, _ = solo[i]
            assert np.array_equal(out[off:off + x.size], x), "batch %d window %d" % (k, j)
            off += x.size
            want_steps += s["num_successful_steps"] + s["num_unsuccessful_steps"]
    assert off == out.size and steps == want_steps


def test_streamed_headline_path_matches_oracle_and_fresh_batches(hip, oracle):
    """The path bench.py's `streamed` number runs, under the oracle: a depth-3 stream of five sets of 1024 windows of
    2000 lines (synth.make_window(0 .. 1023), rotated so that every set is laid out differently) in page-locked arrays, DEFAULT options - so
    the slots take the grouped matrix-core sweep (elimination 4) on the graded cut -3006, the refills are built on the DEVICE
    (csrc/lba_device_build.h: copy-engine ingest, k_build_lines / _rows / _order / _layout / _tiles, k_permute_obs) and the results are written
    in place.  Every window of every set equals, to the byte, the same window in a fresh batch built by the host packer; 16 windows of a
    REFILLED set against oracle.lba_solve: identical accept / reject decisions at every iteration, summaries, and the deviation caps of
    tests/test_gpu_lba.py::test_headline_path_matches_oracle."""
    from test_gpu_lba import TIGHT, _assert_summary_parity, _assert_trace_parity
    B = int(os.environ.get("SLSLAM_HEADLINE_WINDOWS", "1024"))
    ws = [synth.make_window(i, num_lines=2000) for i in range(B)]
    fresh = hip.LBABatch()
    for w in ws:
        fresh.add(w)
    fresh.finalize()
    assert fresh.elimination() == 4
    fresh.solve(); fresh.download()
    want = [fresh.parameters(i).copy() for i in range(B)]
    want_sum = [fresh.summary(i) for i in range(B)]
    cut = fresh.window_chunks(0)
    fresh.close()
    assert cut == -3006
    nsets, depth = 5, 3
    base = hip.WindowSet(ws, pinned=True)
    rots = [(k * 37) % B for k in range(nsets)]
    sets = [base.derive(list(range(r, B)) + list(range(r))) for r in rots]
    st = hip.LBAStream(depth=depth, host_threads=1)
    tickets, res = [], {}
    views = {}
    for k in range(nsets):
        if k >= depth:
            res[k - depth] = st.collect(tickets[k - depth])
            views[k - depth] = None
        tickets.append(st.submit(sets[k]))
    for k in range(nsets - depth, nsets):
        res[k] = st.collect(tickets[k])
    ss, bs = st.stats(), st.build_stats()
    assert ss["builds"] == depth and ss["refills"] == nsets - depth
    assert bs["device_builds"] == nsets - depth and bs["zero_copy"] == nsets - depth and bs["fallback_windows"] == 0
    # the batch that served the last (refilled) set: the headline's sweep and cut, traces for the oracle comparison
    kk = nsets - 1
    bv = st.batch_of(tickets[kk], sets[kk])
    assert bv.elimination() == 4
    # (the windows tests/test_gpu_lba.py::test_headline_path_matches_oracle holds against the oracle, every second one; position j of the set
    # holds window (j + rotation) % B)
    picks_i = sorted(set(int(round(x)) for x in np.linspace(0, B - 1, 32)))[::2]
    picks = [(i - rots[kk]) % B for i in picks_i]
    assert all(bv.window_chunks(j) == -3006 for j in picks)
    traces = {j: bv.trace(j) for j in picks}
    for k in range(nsets):
        r = rots[k]
        for j in range(B):
            i = (j + r) % B                              # window i sits at position j of set k
            assert np.array_equal(sets[k].parameters(j), want[i]), "set %d position %d (window %d)" % (k, j, i)
            assert res[k][j] == want_sum[i]
    CAP = dict(cost=1e-6, radius=5e-5, step_norm=3e-4, rho=1e-3, cam=3e-9, line=1.5e-3)
    beyond, rejected = 0, 0
    for j in picks:
        i = (j + rots[kk]) % B
        w = ws[i]
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = sets[kk].parameters(j), res[kk][j], traces[j]
        assert len(t0) == len(t1)
        for a, c in zip(t0, t1):
            assert a["iteration"] == c["iteration"] and a["step_is_successful"] == c["step_is_successful"] and a["step_is_valid"] == c["step_is_valid"]
        _assert_trace_parity(t0, t1, n=2)
        _assert_summary_parity(s0, s1)
        rejected += s1["num_unsuccessful_steps"]
        nc = 6 * int(w["num_cameras"])
        d = dict(cost=0.0, radius=0.0, step_norm=0.0, rho=0.0)
        for a, c in zip(t0, t1):
            d["cost"] = max(d["cost"], abs(a["cost"] - c["cost"]) / abs(a["cost"]))
            d["radius"] = max(d["radius"], abs(a["trust_region_radius"] - c["trust_region_radius"]) / a["trust_region_radius"])
            d["step_norm"] = max(d["step_norm"], abs(a["step_norm"] - c["step_norm"]) / (a["step_norm"] + 1e-12))
            d["rho"] = max(d["rho"], abs(a["relative_decrease"] - c["relative_decrease"]) / (abs(a["relative_decrease"]) + 1e-3))
        d["cam"] = float(np.abs(x0[:nc] - x1[:nc]).max())
        d["line"] = float(np.abs(x0[nc:] - x1[nc:]).max())
        for q in CAP:
            assert d[q] <= CAP[q], (i, q, d[q])
        beyond += any(d[q] > TIGHT[q] for q in TIGHT)
    assert beyond <= len(picks) // 4 and rejected > 0
    st.close()
    for s_ in sets:
        s_.close()
    base.close()


def test_stream_with_an_oversize_window_among_ordinary_ones(hip, oracle):
    """ A submit that mixes a window beyond the tiled sweeps (24 free cameras: the global-memory path) with ordinary ones is
    solved as a MIXED batch - two parts side by side, the top-level batch holds no LM states of its own: collect must take the step counts and
    summaries through the routing getters.  Two such sets through a depth-2 stream (the second submit to a slot finds a mixed batch: rebuilt),
    every window against a fresh batch of its set, the oversize one also against the oracle."""
    sets = []
    for k in range(3):
        s = [synth.make_window(9100 + 10 * k + i, num_lines=180 + 20 * i) for i in range(4)]
        s.insert(2, synth.make_window(9150 + k, num_lines=50, num_kf=30, num_free=24, mean_track=10.0))
        sets.append(s)
    st = hip.LBAStream(depth=2, host_threads=2)
    wsets = [hip.WindowSet(s) for s in sets]
    tickets, res = [], {}
    for k in range(3):
        if k >= 2:
            res[k - 2] = st.collect(tickets[k - 2])
        tickets.append(st.submit(wsets[k]))
    for k in (1, 2):
        res[k] = st.collect(tickets[k])
    its = st.stats()["lm_iterations"]
    want_its = 0
    for k in range(3):
        fresh, _, _ = _solve_fresh(hip, sets[k])
        for j in range(5):
            assert np.array_equal(wsets[k].parameters(j), fresh[j][0]), (k, j)
            assert res[k][j] == fresh[j][1]
            want_its += fresh[j][1]["num_successful_steps"] + fresh[j][1]["num_unsuccessful_steps"]
    assert its == want_its
    xo, so, _ = oracle.lba_solve(sets[1][2], linear_solver=1)
    assert so["num_successful_steps"] == res[1][2]["num_successful_steps"] and np.abs(xo - wsets[1].parameters(2)).max() < 1e-5
    st.close()


def test_refused_refill_keeps_the_results_of_the_download_under_way(hip):
    """ slslam_lba_batch_refill after download_async: a refill that is REFUSED (windows that cannot fit) leaves the batch as it
    was, the results of the download included - wait() and the getters still deliver them."""
    ws = [synth.make_window(9300 + i, num_lines=200 + 10 * i) for i in range(4)]
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize(refill_headroom_percent=20)
    b.solve(); b.download()
    want = [b.parameters(i).copy() for i in range(4)]
    b.reset(); b.solve(); b.download_async()
    with pytest.raises(hip.SlslamError) as e:
        b.refill([synth.make_window(9400 + i, num_lines=900) for i in range(4)])
    assert e.value.status == 4
    b.wait()
    for i in range(4):
        assert np.array_equal(b.parameters(i), want[i])
    b.close()


def test_a_window_that_fails_numerically_keeps_its_parameters(hip, oracle):
    """Ceres leaves the user's parameters untouched when a solve ends in NUMERICAL_FAILURE (include/slslam_hip.h).  A window whose initial
    cost overflows (one observation at 1e200: finite, so it passes the input tests of every path) ends so at once; its parameter array must
    come back as it went in - through the one-shot call, a fresh batch, a stream of page-locked sets (the in-place export skips the window)
    and a stream of ordinary arrays (the export hands back the initial values) - while its neighbours are solved as if it were not there."""
    sets = []
    for k in range(3):
        s = [synth.make_window(9400 + 10 * k + i, num_lines=200 + 20 * i) for i in range(4)]
        s[1]["observations"] = s[1]["observations"].copy()
        s[1]["observations"].reshape(-1)[8 * 5 + 2] = 1e200
        sets.append(s)
    x, summ, _ = hip.lba_solve(sets[0][1])
    assert summ["termination"] == "NUMERICAL_FAILURE" and np.array_equal(x, sets[0][1]["parameters"])
    _, so, _ = oracle.lba_solve(sets[0][1], linear_solver=1)
    assert so["termination_type"] == summ["termination_type"]
    fresh = [_solve_fresh(hip, s)[0] for s in sets]
    for k in range(3):
        assert fresh[k][1][1]["termination"] == "NUMERICAL_FAILURE" and np.array_equal(fresh[k][1][0], sets[k][1]["parameters"])
        alone = _solve_fresh(hip, [sets[k][0], sets[k][2], sets[k][3]])[0]
        for j, i in enumerate((0, 2, 3)):
            assert fresh[k][i][1]["termination"] != "NUMERICAL_FAILURE"
            assert fresh[k][i][1]["num_successful_steps"] == alone[j][1]["num_successful_steps"]
            assert abs(fresh[k][i][1]["final_cost"] - alone[j][1]["final_cost"]) <= 1e-9 * alone[j][1]["final_cost"]
    for pinned in (True, False):
        st = hip.LBAStream(depth=2, host_threads=2)
        wsets = [hip.WindowSet(s, pinned=pinned) for s in sets]
        tickets, res = [], {}
        for k in range(3):
            if k >= 2:
                res[k - 2] = st.collect(tickets[k - 2])
            tickets.append(st.submit(wsets[k]))
        for k in (1, 2):
            res[k] = st.collect(tickets[k])
        assert st.build_stats()["device_builds"] >= 1
        for k in range(3):
            assert res[k][1]["termination"] == "NUMERICAL_FAILURE"
            assert np.array_equal(wsets[k].parameters(1), sets[k][1]["parameters"]), (pinned, k)
            for i in (0, 2, 3):
                assert np.array_equal(wsets[k].parameters(i), fresh[k][i][0]), (pinned, k, i)
                assert res[k][i] == fresh[k][i][1]
        st.close()
        for ws in wsets:
            ws.close()


@pytest.mark.parametrize("free,kf", [(5, 10), (20, 40)])
def test_stream_at_the_reference_window_sizes(hip, oracle, free, kf):
    """The reference's study runs windows of W = 5 ... 40 keyframes; the bench streams W = 10.  Sets of W = 5 and W = 20 windows
    (free cameras = W, as many constant ones beside them: src/slam.cpp:805-830) through a depth-2 stream of page-locked arrays, built on the
    device from the second round on: every window equal to the byte to the same window in a fresh batch of its set, two of them against the
    oracle (same steps, final cost 1e-7 relative, parameters 1e-5)."""
    per = 8
    sets = [[synth.make_window(9600 + 100 * free + 10 * k + i, num_lines=150 + 25 * i, num_kf=kf, num_free=free, mean_track=0.6 * kf) for i in range(per)]
            for k in range(5)]
    st = hip.LBAStream(depth=2, host_threads=2)
    wsets = [hip.WindowSet(s, pinned=True) for s in sets]
    tickets, res = [], {}
    for k in range(5):
        if k >= 2:
            res[k - 2] = st.collect(tickets[k - 2])
        tickets.append(st.submit(wsets[k]))
    for k in (3, 4):
        res[k] = st.collect(tickets[k])
    bs = st.build_stats()
    assert bs["device_builds"] == 3 and bs["fallback_windows"] == 0, bs
    for k in range(5):
        fresh, _, _ = _solve_fresh(hip, sets[k])
        for j in range(per):
            assert np.array_equal(wsets[k].parameters(j), fresh[j][0]), (free, k, j)
            assert res[k][j] == fresh[j][1]
    for k, j in ((2, 1), (4, 6)):
        xo, so, _ = oracle.lba_solve(sets[k][j], linear_solver=1)
        assert so["num_successful_steps"] == res[k][j]["num_successful_steps"] and so["num_unsuccessful_steps"] == res[k][j]["num_unsuccessful_steps"]
        assert abs(so["final_cost"] - res[k][j]["final_cost"]) <= 1e-7 * so["final_cost"] and np.abs(xo - wsets[k].parameters(j)).max() < 1e-5
    st.close()
    for ws in wsets:
        ws.close()
