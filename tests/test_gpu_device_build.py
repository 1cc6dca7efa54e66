"""GPU tests of the build stage ON THE DEVICE (slslam_amd/csrc/lba_device_build.h, round 6): what LBAProblem::build does per window
(reference src/lba_problem.cpp:54-93, fed by the five arrays of src/slam.cpp:899-921) as four kernels on the arrays as the caller holds
them.  The host packer (lba_pack.cpp) is the specification: everything the device emits is compared with it byte for byte, against
tests/golden/packer_digest.json, and through the solved bytes of refilled batches and streams."""
import importlib.util
import json
import os

import numpy as np
import pytest

from slslam_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ("line_order", "line_ptr", "ob_orig", "ob_cam", "cam_cf", "tiles", "items", "lane_map", "desc")


def _digest_module():
    spec = importlib.util.spec_from_file_location("make_packer_digest", os.path.join(HERE, "golden", "make_packer_digest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _scrambled(seed, **kw):
    rng = np.random.default_rng(seed)
    w = synth.make_window(seed, **kw)
    perm = rng.permutation(len(w["camera_index"]))
    for k in ("camera_index", "line_index"):
        w[k] = np.asarray(w[k])[perm]
    w["observations"] = np.asarray(w["observations"]).reshape(-1, 8)[perm].reshape(-1)
    fx = np.asarray(w["fixed_index"]).reshape(-1, 2)[perm].copy()
    const = rng.random(w["num_lines"]) < 0.2
    fx[:, 1] = const[w["line_index"]]
    w["fixed_index"] = fx.reshape(-1)
    return w


def _compare(hip, host_math, w, grouping, name):
    from test_host_side import _pack
    rc, P = _pack(host_math, w, grouping=grouping)
    assert rc == 0, name
    st, D = hip.debug_device_pack(w, grouping=grouping)
    assert st == 0, (name, st)
    for k in ("Cf", "ntiles", "nitems", "nfree", "nkept"):
        assert D[k] == P[k], (name, grouping, k, D[k], P[k])
    for k in FIELDS:
        a, b = np.asarray(D[k]), np.asarray(P[k])
        assert a.shape == b.shape and np.array_equal(a, b), (name, grouping, k, int(np.argmax(a.reshape(-1) != b.reshape(-1))) if a.shape == b.shape else (a.shape, b.shape))
    return D


def test_device_build_reproduces_the_packer_digest(hip, host_math):
    """The family of tests/golden/packer_digest.json (bench shapes, tracks of 40 keyframes, 20 free cameras, motion-only, scrambled order
    with holes and constant lines), both packings: every array the device build emits equals the host packer's, and its crc is the
    golden one - the packed LAYOUT, of which a window's solved bytes are a function, does not depend on who built it."""
    mod = _digest_module()
    gold = json.load(open(os.path.join(HERE, "golden", "packer_digest.json")))
    seen = 0
    for name, w in mod.family().items():
        for g in (0, 1):
            D = _compare(hip, host_math, w, g, name)
            ref = gold["%s/grouping%d" % (name, g)]
            assert (D["ntiles"], D["nitems"]) == (ref["tiles"], ref["items"]), (name, g)
            assert mod.digest(D) == ref["crc32"], (name, g)
            seen += 1
    assert seen == len(gold) == 12


def test_build_tiles_without_lds_staging(hip, host_math, monkeypatch):
    """k_build_tiles keeps the tile loop's inputs in LDS when a window's observations fit; the form that reads them in place (windows
    beyond that) is the same bytes: part of the digest family and a scrambled window through SLSLAM_BUILD_TILES_UNSTAGED."""
    monkeypatch.setenv("SLSLAM_BUILD_TILES_UNSTAGED", "1")
    mod = _digest_module()
    gold = json.load(open(os.path.join(HERE, "golden", "packer_digest.json")))
    for name, w in list(mod.family().items())[:3]:
        for g in (0, 1):
            D = _compare(hip, host_math, w, g, name)
            assert mod.digest(D) == gold["%s/grouping%d" % (name, g)]["crc32"], (name, g)
    for g in (0, 1):
        _compare(hip, host_math, _scrambled(58, num_lines=300), g, "scrambled58")


def test_device_build_equals_host_packer_on_varied_windows(hip, host_math):
    """Shapes the digest family does not hold: few and many lines, every track length (long lines take whole rows, short ones need several
    sin / cos rounds), lines nobody observes, constant cameras among the free ones, scrambled caller order, one camera, no observation."""
    cases = []
    for seed, kw in ((31, dict(num_lines=1)), (32, dict(num_lines=7, num_kf=4, num_free=2)), (33, dict(num_lines=64, mean_track=1.5)),
                     (34, dict(num_lines=700, mean_track=3.0)), (35, dict(num_lines=120, num_kf=40, num_free=20, mean_track=30.0)),
                     (36, dict(num_lines=90, num_kf=64, num_free=10, mean_track=50.0)), (37, dict(num_lines=1500)),
                     (38, dict(num_lines=333, num_kf=14, num_free=9)), (39, dict(num_lines=2000))):
        cases.append(("synth%d" % seed, synth.make_window(seed, **kw)))
    for seed, kw in ((51, dict(num_lines=400)), (52, dict(num_lines=250, mean_track=2.0)), (53, dict(num_lines=180, num_kf=30, num_free=15, mean_track=20.0))):
        cases.append(("scrambled%d" % seed, _scrambled(seed, **kw)))
    w = synth.make_window(61, num_lines=150)
    w = dict(w, num_lines=w["num_lines"] + 9, parameters=np.concatenate([w["parameters"], np.tile([0.1, 0.2, 0.3, 0.4], 9)]))   # nine lines without observations
    cases.append(("unobserved_lines", w))
    w = synth.make_window(62, num_lines=140)
    fx = np.asarray(w["fixed_index"]).reshape(-1, 2).copy()
    fx[np.asarray(w["camera_index"]) == 3, 0] = 1                       # a free camera made constant by its flags
    cases.append(("constant_camera", dict(w, fixed_index=fx.reshape(-1))))
    cases.append(("motion_only", synth.make_motion_only(63, num_lines=57)))
    w = synth.make_window(64, num_lines=20)
    cases.append(("no_observations", dict(w, camera_index=np.zeros(0, np.int32), line_index=np.zeros(0, np.int32), fixed_index=np.zeros(0, np.int32),
                                         observations=np.zeros(0))))
    for name, w in cases:
        for g in (0, 1):
            _compare(hip, host_math, w, g, name)


def test_device_build_flags_what_it_leaves_to_the_host_path(hip):
    """Bad input (an index out of range, NaN in the observations or the parameters) is flagged 1; shapes the tiled device build does not take -
    a camera that sees a line twice, a line with more than 64 observations, more than 20 free cameras - are flagged 2 (the host path solves
    them: test_stream_hands_flagged_windows_to_the_host_path)."""
    w = synth.make_window(71, num_lines=100)
    cam = np.asarray(w["camera_index"]).copy(); cam[5] = w["num_cameras"]
    assert hip.debug_device_pack(dict(w, camera_index=cam))[0] == 1
    line = np.asarray(w["line_index"]).copy(); line[7] = -1
    assert hip.debug_device_pack(dict(w, line_index=line))[0] == 1
    obs = np.asarray(w["observations"], dtype=np.float64).copy().reshape(-1); obs[11] = np.nan
    assert hip.debug_device_pack(dict(w, observations=obs))[0] == 1
    prm = np.asarray(w["parameters"], dtype=np.float64).copy(); prm[-3] = np.inf
    assert hip.debug_device_pack(dict(w, parameters=prm))[0] == 1
    cam = np.asarray(w["camera_index"]).copy()
    same = np.flatnonzero(np.asarray(w["line_index"]) == w["line_index"][0])
    cam[same[1]] = cam[same[0]]                                          # camera sees line twice
    assert hip.debug_device_pack(dict(w, camera_index=cam))[0] == 2
    assert hip.debug_device_pack(synth.make_window(72, num_lines=40, num_kf=30, num_free=24, mean_track=10.0))[0] == 2      # 24 free cameras
    with pytest.raises(hip.SlslamError) as e:                            # 80 cameras: refused before anything is launched
        hip.debug_device_pack(synth.make_window(31, num_lines=100, num_kf=80, num_free=40, mean_track=30.0))
    assert e.value.status == 4
    assert hip.debug_device_pack(w)[0] == 0


def _solve_fresh(hip, ws, **opt):
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize(**opt)
    b.solve(); b.download()
    out = [(b.parameters(i).copy(), b.summary(i), b.trace(i)) for i in range(len(ws))]
    cuts = [b.window_chunks(i) for i in range(len(ws))]
    b.close()
    return out, cuts


@pytest.mark.parametrize("elim", [1, 4])
@pytest.mark.parametrize("pinned", [False, True])
def test_device_built_refill_equals_fresh_batch(hip, oracle, elim, pinned):
    """slslam_lba_batch_refill with the build stage on the device - from page-locked arrays the GPU reads in place, and from ordinary arrays
    through the staging copy - against fresh batches built by the host packer: parameters, summaries, iteration traces and chunk cuts
    identical to the byte over three sets of windows of differing sizes; a refill built by the host packer (device_build = -1) gives the
    same bytes again; the first refilled set also against the oracle."""
    sets = [[synth.make_window(100 * k + i, num_lines=n) for i, n in enumerate((300, 420, 380, 350, 400, 330))] for k in range(3)]
    sets[1][2] = _scrambled(177, num_lines=390)
    b = hip.LBABatch()
    for w in sets[0]:
        b.add(w)
    b.finalize(lba_elimination=elim, refill_headroom_percent=25, host_threads=2)
    keep = []
    for k in (1, 2, 0, 1):
        ws = hip.WindowSet(sets[k], pinned=pinned)
        keep.append(ws)
        b.refill(ws)
        b.solve(); b.download()
        fresh, cuts = _solve_fresh(hip, sets[k], lba_elimination=elim)
        for i in range(len(sets[k])):
            assert b.window_chunks(i) == cuts[i]
            assert np.array_equal(b.parameters(i), fresh[i][0]), "set %d window %d" % (k, i)
            assert b.summary(i) == fresh[i][1] and b.trace(i) == fresh[i][2]
        b.reset(); b.solve(); b.download()                  # a refilled batch is reset from ITS windows' initial values
        for i in range(len(sets[k])):
            assert np.array_equal(b.parameters(i), fresh[i][0])
    for i in (0, 3):
        xo, so, _ = oracle.lba_solve(sets[1][i], linear_solver=1)
        assert so["num_successful_steps"] == b.summary(i)["num_successful_steps"]
        assert abs(so["final_cost"] - b.summary(i)["final_cost"]) <= 1e-7 * so["final_cost"] and np.abs(xo - b.parameters(i)).max() < 1e-5
    too_big = [synth.make_window(900 + i, num_lines=900) for i in range(6)]
    with pytest.raises(hip.SlslamError) as e:
        b.refill(too_big)                                   # sizes the host knows cannot fit: refused at once, batch unchanged
    assert e.value.status == 4
    b.reset(); b.solve(); b.download()
    fresh, _ = _solve_fresh(hip, sets[1], lba_elimination=elim)
    for i in range(6):
        assert np.array_equal(b.parameters(i), fresh[i][0])
    b.close()
    for ws in keep:
        ws.close()


def test_stream_hands_flagged_windows_to_the_host_path(hip, oracle):
    """A stream whose sets hold windows the device build flags - a camera that sees a line twice (the reference's map never does, the ABI does
    not forbid it) - among ordinary ones: the flagged windows are solved through the host path at collect time, the others by the batch;
    every window equals what slslam_lba_solve / a fresh batch returns; stats count the device builds and the fallbacks.  Bad input: an
    index out of range is refused at submit time (the host threads narrow the indices); a NaN the device alone sees is reported by collect
    (SLSLAM_ERR_INVALID_ARGUMENT) and the other windows of the set are solved all the same."""
    per = 6
    base = [[synth.make_window(7000 + 10 * k + i, num_lines=220 + 15 * i) for i in range(per)] for k in range(5)]
    dup = dict(base[3][2])
    cam = np.asarray(dup["camera_index"]).copy()
    same = np.flatnonzero(np.asarray(dup["line_index"]) == dup["line_index"][0])
    cam[same[1]] = cam[same[0]]
    dup["camera_index"] = cam
    base[3][2] = dup
    for pinned in (True, False):
        st = hip.LBAStream(depth=2, host_threads=2)
        wsets = [hip.WindowSet(s, pinned=pinned) for s in base]
        tickets, res = [], {}
        for k in range(5):
            if k >= 2:
                res[k - 2] = st.collect(tickets[k - 2])
            tickets.append(st.submit(wsets[k]))
        for k in (3, 4):
            res[k] = st.collect(tickets[k])
        bs, ss = st.build_stats(), st.stats()
        assert ss["builds"] == 2 and ss["refills"] == 3
        assert bs["device_builds"] == 3 and bs["fallback_windows"] == 1 and bs["zero_copy"] == (3 if pinned else 0)
        for k in range(5):
            fresh, _ = _solve_fresh(hip, [w for j, w in enumerate(base[k]) if not (k == 3 and j == 2)])
            fi = 0
            for j in range(per):
                if k == 3 and j == 2:
                    x, s, _ = hip.lba_solve(base[k][j])
                    assert np.array_equal(wsets[k].parameters(j), x) and res[k][j] == s
                    continue
                assert np.array_equal(wsets[k].parameters(j), fresh[fi][0]), (pinned, k, j)
                assert res[k][j] == fresh[fi][1]
                fi += 1
        # bad input in a refill.  An index out of range is found by whoever narrows the indices - the host threads, at submit time, as the
        # host packer would; a NaN among page-locked observations only the device sees: flagged there, reported by collect, the rest of the set solved
        bad = [dict(w) for w in base[1]]
        cam = np.asarray(bad[4]["camera_index"]).copy(); cam[3] = 99
        bad[4]["camera_index"] = cam
        wb = hip.WindowSet(bad, pinned=pinned)
        with pytest.raises(hip.SlslamError) as e:
            st.submit(wb)
        assert e.value.status == 1
        wb.close()
        bad = [dict(w) for w in base[1]]
        ob = np.asarray(bad[4]["observations"], dtype=np.float64).reshape(-1).copy(); ob[17] = np.nan
        bad[4]["observations"] = ob
        wb = hip.WindowSet(bad, pinned=pinned)
        if pinned:
            t = st.submit(wb)
            with pytest.raises(hip.SlslamError) as e:
                st.collect(t)
            assert e.value.status == 1
            fresh, _ = _solve_fresh(hip, base[1])
            for j in (0, 1, 2, 3, 5):
                assert np.array_equal(wb.parameters(j), fresh[j][0])
            assert np.array_equal(wb.parameters(4), np.asarray(bad[4]["parameters"]))        # untouched
        else:
            with pytest.raises(hip.SlslamError) as e:
                st.submit(wb)                               # the staging copy tests the values on the host, as the host packer does
            assert e.value.status == 1
        st.close()
        for ws in wsets:
            ws.close()
        wb.close()


@pytest.mark.parametrize("mode", ["pageable", "pinned", "packed"])
def test_stream_rebuilds_a_slot_whose_refill_did_not_fit(hip, mode):
    """Only the device knows how many tiles a set needs.  A set with fewer lines and observations than the slot was built for, but long
    tracks (a line of twenty observations takes two rows of 16 lanes, four short ones share one), passes the host's size tests and is flagged by k_build_layout as a whole:
    collect packs it on the host threads into a batch of its own, solves it as ONE batch - what a fresh batch of the set returns, to the
    byte - and that batch takes the slot, so that the next set of the shape is built on the device."""
    per = 40
    sets = [[synth.make_window(7700 + i, num_lines=300, num_kf=40, num_free=10, mean_track=5.0) for i in range(per)]]      # 28.9 k observations, 494 tiles per 24 windows
    for k in (1, 2):
        sets.append([synth.make_window(7800 + 100 * k + i, num_lines=60, num_kf=40, num_free=10, mean_track=36.0) for i in range(per)])     # 24.7 k, 568
    assert sum(len(w["camera_index"]) for w in sets[1]) < sum(len(w["camera_index"]) for w in sets[0])
    st = hip.LBAStream(depth=1, host_threads=2, refill_headroom_percent=5)     # (set 1 needs 15 % more tiles than set 0, set 2 4 % more pair items than set 1)
    wsets = [hip.WindowSet(s, pinned=mode != "pageable", packed=mode == "packed") for s in sets]
    res = [st.collect(st.submit(ws)) for ws in wsets]
    ss, bs = st.stats(), st.build_stats()
    assert ss["builds"] == 2 and ss["refills"] == 2, ss                     # set 1: refill accepted by the host, rebuilt at collect; set 2: a refill that fits
    assert bs["device_builds"] == 2 and bs["fallback_windows"] == per, bs
    for k in range(3):
        fresh, _ = _solve_fresh(hip, sets[k])
        for j in range(per):
            assert np.array_equal(wsets[k].parameters(j), fresh[j][0]), (mode, k, j)
            assert res[k][j] == fresh[j][1]
    st.close()
    for ws in wsets:
        ws.close()


def test_stream_with_narrowed_indices(hip):
    """slslam_lba_stream_submit_packed: the caller hands the three index arrays of every window narrowed to one 32-bit word per observation
    (slslam_pack_indices) - from page-locked memory (read in place) and from ordinary memory (staging copy).  Same bytes as fresh batches;
    the slot's first batches (host packer) get the arrays expanded again; a word that names a camera the window does not have is refused."""
    per = 5
    sets = [[synth.make_window(8000 + 10 * k + i, num_lines=200 + 20 * i) for i in range(per)] for k in range(4)]
    for pinned in (True, False):
        st = hip.LBAStream(depth=2, host_threads=2)
        wsets = [hip.WindowSet(s, pinned=pinned, packed=True) for s in sets]
        tickets, res = [], {}
        for k in range(4):
            if k >= 2:
                res[k - 2] = st.collect(tickets[k - 2])
            tickets.append(st.submit(wsets[k]))
        for k in (2, 3):
            res[k] = st.collect(tickets[k])
        bs = st.build_stats()
        assert bs["device_builds"] == 2 and bs["fallback_windows"] == 0
        for k in range(4):
            fresh, _ = _solve_fresh(hip, sets[k])
            for j in range(per):
                assert np.array_equal(wsets[k].parameters(j), fresh[j][0]), (pinned, k, j)
                assert res[k][j] == fresh[j][1]
        bad = hip.WindowSet(sets[1], pinned=pinned, packed=True)
        bad.packed_arrays[2][7] = np.uint32(int(bad.packed_arrays[2][7]) | (200 << 16))         # camera 200+
        if pinned:
            t = st.submit(bad)
            with pytest.raises(hip.SlslamError) as e:
                st.collect(t)
            assert e.value.status == 1
        else:
            with pytest.raises(hip.SlslamError) as e:
                st.submit(bad)
            assert e.value.status == 1
        st.close()
        for ws in wsets + [bad]:
            ws.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_device_built_refill_edge_shapes(hip, oracle, pinned):
    """Refills whose windows are degenerate: a window without observations, one whose blocks are all constant, unused camera / line slots with a
    scrambled order, a motion-only shaped window among general ones, fewer cameras than the slot was made for - and a set whose EVERY window is
    flagged (all emitted empty: nothing is launched, every result comes from the host path).  Same bytes as fresh batches."""
    base = [synth.make_window(9500 + i, num_lines=150 + 10 * i) for i in range(5)]
    w = synth.make_window(9510, num_lines=60)
    rng = np.random.default_rng(11)
    perm = rng.permutation(len(w["camera_index"]))
    ragged = dict(w, num_cameras=22, num_lines=63, camera_index=w["camera_index"][perm], line_index=w["line_index"][perm],
                  fixed_index=w["fixed_index"].reshape(-1, 2)[perm].reshape(-1), observations=w["observations"].reshape(-1, 8)[perm].reshape(-1),
                  parameters=np.concatenate([w["parameters"][:120], rng.normal(size=12), w["parameters"][120:], rng.uniform(0.2, 1, 12)]))
    empty = dict(num_cameras=2, num_lines=3, camera_index=np.zeros(0, np.int32), line_index=np.zeros(0, np.int32), fixed_index=np.zeros(0, np.int32),
                 observations=np.zeros(0), parameters=np.arange(24.0))
    wc = synth.make_window(9511, num_lines=40)
    all_const = dict(wc, fixed_index=np.ones_like(wc["fixed_index"]))
    few = synth.make_window(9512, num_lines=80, num_kf=8, num_free=3)
    other = [ragged, empty, all_const, few, synth.make_window(9513, num_lines=170)]
    # (the slot's first batch sets the room: 22 cameras in a window, 10 free, twice the lines)
    first = [dict(ragged), synth.make_window(9520, num_lines=300), synth.make_window(9521, num_lines=320), synth.make_window(9522, num_lines=280),
             synth.make_window(9523, num_lines=310)]
    st = hip.LBAStream(depth=1, host_threads=2, refill_headroom_percent=50)
    for k, s in enumerate((first, base, other, base)):
        ws = hip.WindowSet(s, pinned=pinned)
        res = st.collect(st.submit(ws))
        fresh, _ = _solve_fresh(hip, s)
        for j in range(5):
            assert np.array_equal(ws.parameters(j), fresh[j][0]), (k, j)
            assert res[j] == fresh[j][1], (k, j)
        ws.close()
    assert st.build_stats()["device_builds"] == 3 and st.build_stats()["fallback_windows"] == 0
    # every window flagged: cameras that see a line twice in all five
    dups = []
    for wd_ in base:
        cam = np.asarray(wd_["camera_index"]).copy()
        same = np.flatnonzero(np.asarray(wd_["line_index"]) == wd_["line_index"][0])
        cam[same[1]] = cam[same[0]]
        dups.append(dict(wd_, camera_index=cam))
    ws = hip.WindowSet(dups, pinned=pinned)
    res = st.collect(st.submit(ws))
    assert st.build_stats()["fallback_windows"] == 5
    for j in range(5):
        x, s_, _ = hip.lba_solve(dups[j])
        assert np.array_equal(ws.parameters(j), x) and res[j] == s_
    ws.close()
    st.close()
