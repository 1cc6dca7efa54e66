"""CPU tests of the product's host-side code: the device math header compiled for the host,
the window packer, and the C ABI surface (no compute calls without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from slslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


def _dp(a):
    return a.ctypes.data_as(DP)


def _ip(a):
    return a.ctypes.data_as(IP)


def test_analytic_jacobian_matches_oracle_dual_numbers(host_math, oracle):
    """slslam_amd/csrc/lba_math.h (what every LBA kernel evaluates) vs the oracle's Jet<10>
    restatement of AutoDiffCostFunction<LineReprojectionError,4,6,4> (lba_problem.cpp:65-74)."""
    w = synth.make_window(3, num_lines=200)
    prm, Cn = w["parameters"], w["num_cameras"]
    worst = np.zeros(3)
    for i in range(len(w["camera_index"])):
        cam = prm[6 * w["camera_index"][i]:][:6].copy()
        line = prm[6 * Cn + 4 * w["line_index"][i]:][:4].copy()
        ob = w["observations"][i].copy()
        r0, jc0, jl0 = oracle.line_residual_jet(cam, line, ob)
        r, jc, jl, r2 = np.zeros(4), np.zeros((4, 6)), np.zeros((4, 4)), np.zeros(4)
        host_math.hm_obs_linearise(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(r), _dp(jc), _dp(jl))
        host_math.hm_obs_residual(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(r2))
        worst = np.maximum(worst, [max(abs(r - r0).max(), abs(r2 - r0).max()),
                                   (abs(jc - jc0) / (1 + abs(jc0))).max(), (abs(jl - jl0) / (1 + abs(jl0))).max()])
    assert worst[0] < 1e-14 and worst[1] < 1e-13 and worst[2] < 1e-13


def test_gram_formulation_matches_explicit_jacobians(host_math):
    """lba_gram.h (what the matrix-core elimination sweep evaluates: W = sum g g^T per observation, every block a product
    with it) against the explicit analytic Jacobians of lba_math.h.  Raw camera coordinates: J_c = J_c' T_c with
    T_c = diag(JL, I) — the reduced solve applies T_c — so J_c' is recovered from J_c with JL^-1."""
    rng = np.random.default_rng(11)
    w = synth.make_window(6, num_lines=60)
    Cn, prm = w["num_cameras"], w["parameters"]
    worst = 0.0
    for i in range(0, len(w["camera_index"]), 2):
        cam = prm[6 * w["camera_index"][i]:][:6].copy()
        if i % 6 == 0:
            cam[:3] = 0.0                                    # the identity keyframe: first-order branch of the rotation
        line = prm[6 * Cn + 4 * w["line_index"][i]:][:4].copy()
        ob = w["observations"][i].copy()
        sl = rng.uniform(0.2, 1.5, size=4)
        r, jc, jl = np.zeros(4), np.zeros((4, 6)), np.zeros((4, 4))
        host_math.hm_obs_linearise(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(r), _dp(jc), _dp(jl))
        R, JL = np.zeros(9), np.zeros(9)
        host_math.hm_cam_prepare(_dp(cam), _dp(R), _dp(JL))
        jcp = jc.copy()
        jcp[:, :3] = jc[:, :3] @ np.linalg.inv(JL.reshape(3, 3))       # rows tau^T
        jls = jl * sl
        r2, D, gc, G, H, gl = np.zeros(4), np.zeros(21), np.zeros(6), np.zeros((6, 4)), np.zeros(10), np.zeros(4)
        host_math.hm_obs_gram_blocks(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(sl), _dp(r2), _dp(D), _dp(gc), _dp(G),
                                     _dp(H), _dp(gl))
        Dw, Hw = jcp.T @ jcp, jls.T @ jls
        tri6 = np.array([Dw[a, b] for a in range(6) for b in range(a + 1)])
        tri4 = np.array([Hw[a, b] for a in range(4) for b in range(a + 1)])
        for got, want in ((r2, r), (D, tri6), (gc, jcp.T @ r), (G, jcp.T @ jls), (H, tri4), (gl, jls.T @ r)):
            worst = max(worst, abs(got - want).max() / (1e-300 + abs(want).max()))
    assert worst < 5e-12, worst


def test_mfma_elimination_index_maps_replay_a_window(host_math):
    """The matrix-core elimination sweep (lba_eliminate_mfma.h) replayed on the CPU through the index maps the kernels use
    (lba_eliminate_mfma_maps.h compiled for the host): every observation of a free camera stores a random 6x4 F block in
    the panel slab of its lane; per line, lane l fetches X[16 I + (l & 15)][l >> 4] for the blocks the line touches and the
    touched accumulator tiles take the rank-4 update X_I X_J^T with the v_mfma_f64_16x16x4_f64 register layout; decoded
    with acc_row_col (what k_reduced_solve does) the tiles must hold sum_lines sum_{i,j} F_i F_j^T."""
    rng = np.random.default_rng(5)
    w = synth.make_window(9, num_lines=150)
    rc, P = _pack(host_math, w)
    assert rc == 0
    Cn, L, M = w["num_cameras"], w["num_lines"], len(w["camera_index"])
    cam = np.ascontiguousarray(w["camera_index"], dtype=np.int32)
    line = np.ascontiguousarray(w["line_index"], dtype=np.int32)
    fixed = np.ascontiguousarray(w["fixed_index"], dtype=np.int32).reshape(-1)
    obs = np.ascontiguousarray(w["observations"], dtype=np.float64).reshape(-1)
    prm = np.array(w["parameters"], dtype=np.float64)
    desc, dup = np.zeros(L, dtype=np.uint32), C.c_int(-1)
    assert host_math.hm_line_desc(Cn, L, M, _ip(cam), _ip(line), _ip(fixed), _dp(obs), _dp(prm),
                                  desc.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(dup)) == 0
    assert dup.value == 0
    ncf = P["Cf"]
    n = 6 * ncf
    want = np.zeros((64, 64))
    tiles = np.zeros((10, 4, 64))                        # accumulator registers [tile][q][lane]
    bm = [host_math.hm_block_cam_mask(I) for I in range(4)]
    nmfma = 0
    for t, (lb, nl, flags, ni) in enumerate(P["tiles"]):
        panel = np.full(4 * 400 + 64, np.nan)
        lane_slot, lane_pos = P["lane_map"][t] & 0xFF, (P["lane_map"][t] >> 8) & 0x3F
        F = {}
        for lane in range(64):
            if lane_slot[lane] == 0xFF:
                continue
            s_ = lb + int(lane_slot[lane])
            k = int(P["line_ptr"][s_ + 1] - P["line_ptr"][s_])
            if lane_pos[lane] >= k:
                continue
            cf = P["cam_cf"][P["ob_cam"][P["line_ptr"][s_] + int(lane_pos[lane])]]
            if cf < 0:
                continue
            Fb = rng.normal(size=(6, 4))
            F[(s_, int(cf))] = Fb
            for a in range(6):
                for k4 in range(4):
                    panel[host_math.hm_panel_store(lane, a, k4)] = Fb[a, k4]
        for q in range(nl):
            s_ = lb + q
            mask, first, tbits = int(desc[s_]) & 0x3FF, (int(desc[s_]) >> 10) & 63, int(desc[s_]) >> 16
            cams = [c for c in range(ncf) if (mask >> c) & 1]
            assert sorted(c for (ss, c) in F if ss == s_) == cams          # the mask names the line's free cameras
            assert first == int(np.nonzero(lane_slot == q)[0][0])
            X = np.zeros((4, 64))
            for I in range(4):
                if not (mask & bm[I]):
                    continue
                for lane in range(64):
                    idx = host_math.hm_panel_fetch(lane, I, mask, first)
                    if idx >= 0:
                        X[I, lane] = panel[idx]
            assert not np.isnan(X).any()
            stack = np.zeros((64, 4))
            for c in cams:
                stack[6 * c:6 * c + 6] = F[(s_, c)]
            want += stack @ stack.T
            for tt in range(10):
                I, J = (3 if tt >= 6 else 2 if tt >= 3 else 1 if tt >= 1 else 0), 0
                J = tt - I * (I + 1) // 2
                assert bool((mask & bm[I]) and (mask & bm[J])) == bool((tbits >> tt) & 1)     # the packer's tile bits
                if (tbits >> tt) & 1:
                    nmfma += 1
                    Am = np.array([[X[I, m + 16 * kk] for kk in range(4)] for m in range(16)])     # A[m][k] = a(lane = m + 16 k)
                    Bm = np.array([[X[J, nn + 16 * kk] for nn in range(16)] for kk in range(4)])   # B[k][n] = b(lane = n + 16 k)
                    Dm = Am @ Bm
                    for lane in range(64):
                        for qq in range(4):
                            tiles[tt, qq, lane] += Dm[(lane >> 4) + 4 * qq, lane & 15]
    got = np.zeros((64, 64))
    row, col = C.c_int(0), C.c_int(0)
    for tt in range(10):
        for qq in range(4):
            for lane in range(64):
                host_math.hm_acc_rc(tt, qq, lane, C.byref(row), C.byref(col))
                got[row.value, col.value] = tiles[tt, qq, lane]
    lower = np.tril(np.ones((64, 64), dtype=bool))
    assert abs(got - want)[lower].max() < 1e-11 and abs(want[n:, :]).max() == 0.0
    assert 3.0 < nmfma / L < 6.0                          # ~4.4 MFMA per line on sliding-window visibility
    # both waves of a two-wave workgroup together own every tile exactly once
    owned = sorted(host_math.hm_ptile_of(2, wv, e) for wv in range(2) for e in range(5))
    assert owned == list(range(10)) and [host_math.hm_ptile_of(1, 0, e) for e in range(10)] == list(range(10))


def _hole_window(seed):
    """A window whose lines are seen by arbitrary SUBSETS of the free cameras (camera ranges with holes), some constant lines,
    some lines no free camera sees, observations in scrambled order."""
    rng = np.random.default_rng(seed)
    Cn, ncf, L = 14, 10, 90
    cam, line = [], []
    for l in range(L):
        k = int(rng.integers(1, 11))
        cs = list(rng.choice(ncf, size=k, replace=False)) if l % 9 else []          # every ninth line: fixed cameras only
        cs += list(ncf + rng.choice(Cn - ncf, size=int(rng.integers(0, 4)), replace=False))
        if not cs: cs = [ncf]
        cam += cs; line += [l] * len(cs)
    cam, line = np.asarray(cam, dtype=np.int32), np.asarray(line, dtype=np.int32)
    perm = rng.permutation(len(cam))
    cam, line = cam[perm], line[perm]
    fixed = np.zeros((len(cam), 2), dtype=np.int32)
    fixed[cam >= ncf, 0] = 1
    fixed[line % 7 == 3, 1] = 1
    return dict(num_cameras=Cn, num_lines=L, camera_index=cam, line_index=line, fixed_index=fixed.reshape(-1),
                observations=rng.normal(size=(len(cam), 8)), parameters=rng.normal(size=6 * Cn + 4 * L))


@pytest.mark.parametrize("which", ["sliding", "wide", "holes"])
def test_grouped_elimination_maps_replay_windows(host_math, which):
    """The grouped matrix-core elimination sweep (lba_eliminate_grouped.h) replayed on the CPU through the index maps the kernel
    and the packer use (lba_eliminate_grouped_maps.h compiled for the host).  Packing with grouping = 1 keeps every invariant of the
    default packing (_check_tiles) and adds: a tile's descriptors come group after group (first free camera), inside a group by
    their number of 16-row blocks, lines without elimination work last; every field of a descriptor is what the line's
    observations say.  Then the sweep itself: every observation of a free camera stores a random 6 x 4 F block in its lane's panel
    slab; per descriptor, lane l fetches X[16 r + (l & 15)][l >> 4] (gp_fetch_index) and the group-local tiles take the rank-4
    update with the v_mfma_f64_16x16x4_f64 register layout; the tiles are added into the slab (gp_flush_index) whenever the
    group changes, the fourth block row after every tile of lines.  Decoded with acc_row_col (what k_reduced_solve does) the slab
    must hold sum_lines sum_{i,j} F_i F_j^T."""
    rng = np.random.default_rng(11)
    w = {"sliding": lambda: synth.make_window(9, num_lines=150),
         "wide": lambda: synth.make_window(10, num_lines=60, num_kf=24, num_free=10, mean_track=40.0),
         "holes": lambda: _hole_window(4)}[which]()
    rc0, P0 = _pack(host_math, w)
    rc, P = _pack(host_math, w, grouping=1)
    assert rc0 == 0 and rc == 0
    L, M = int(w["num_lines"]), len(w["camera_index"])
    assert sorted(P["line_order"]) == list(range(L)) and sorted(P["ob_orig"]) == list(range(M))
    for k in ("Cf", "nfree", "nkept"):
        assert P[k] == P0[k]
    assert P["nitems"] == 0 and P0["nitems"] > 0           # the grouped sweep has no pair phase: no work items are built for it (round 5)
    assert np.array_equal(P["cam_cf"], P0["cam_cf"])
    assert P["ntiles"] <= P0["ntiles"] + max(2, P0["ntiles"] // 20)          # grouping costs (almost) no lanes
    _check_tiles(P, w, L, with_items=False)
    ncf = P["Cf"]
    n = 6 * ncf
    fixed_line = np.zeros(L, dtype=bool)
    fixed_line[np.asarray(w["line_index"])[np.asarray(w["fixed_index"]).reshape(-1, 2)[:, 1] != 0]] = True
    slab = np.zeros(10 * 256)
    want = np.zeros((64, 64))
    acc = np.zeros((6, 4, 64))
    cur_a, nflush, nmfma, groups_seen = -1, 0, 0, []

    def flush(tiles, rcs, a):
        for (r, c), T in zip(rcs, tiles):
            for q in range(4):
                for lane in range(64):
                    idx = host_math.hm_gp_flush(r, c, a, q, lane, n)
                    if idx >= 0:
                        slab[idx] += T[q, lane]
            T[:] = 0.0
    RC6 = [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2)]

    def product(Xr, Xc):
        Am = np.array([[Xr[m + 16 * kk] for kk in range(4)] for m in range(16)])      # A[m][k] = a(lane = m + 16 k)
        Bm = np.array([[Xc[nn + 16 * kk] for nn in range(16)] for kk in range(4)])    # B[k][n] = b(lane = n + 16 k)
        Dm = Am @ Bm
        return np.array([[Dm[(lane >> 4) + 4 * qq, lane & 15] for lane in range(64)] for qq in range(4)])
    for t, (lb, nl, flags, ni) in enumerate(P["tiles"]):
        panel = np.full(host_math.hm_gp_panel_doubles(), np.nan)
        panel[64 * 26:] = 0.0                                   # the zero slab
        lane_slot, lane_pos = P["lane_map"][t] & 0xFF, (P["lane_map"][t] >> 8) & 0x3F
        F, first_of, mask_of = {}, {}, {}
        for q in range(nl):
            s_ = lb + q
            lanes = np.nonzero(lane_slot == q)[0]
            first_of[int(lanes[0])] = s_
            sel = P["ob_orig"][P["line_ptr"][s_]:P["line_ptr"][s_ + 1]]
            cfs = P["cam_cf"][np.asarray(w["camera_index"])[sel]]
            mask_of[s_] = 0
            for jj, cf in enumerate(cfs):
                lane = int(lanes[0]) + jj
                Fb = rng.normal(size=(6, 4))                    # (fixed-camera lanes store blocks too: never fetched)
                for a in range(6):
                    for k4 in range(4):
                        panel[host_math.hm_gp_store(lane, a, k4)] = Fb[a, k4]
                if cf >= 0 and not fixed_line[P["line_order"][s_]]:
                    F[(s_, int(cf))] = Fb
                    mask_of[s_] |= 1 << int(cf)
        descs = [int(d) for d in P["desc"][lb:lb + nl]]
        keys = [(((d >> 16) & 15) * 8 + ((d >> 20) & 7)) if d & 0x3FF else 1 << 20 for d in descs]
        assert keys == sorted(keys)                             # group after group, by block count, idle lines last
        seen_lines = set()
        row3 = np.zeros((4, 4, 64))
        row3_used = False
        for d in descs:
            mask, first, a, nb, holes, wdt = d & 0x3FF, (d >> 10) & 63, (d >> 16) & 15, (d >> 20) & 7, (d >> 23) & 1, (d >> 24) & 15
            s_ = first_of[first]
            assert s_ not in seen_lines
            seen_lines.add(s_)
            assert mask == mask_of[s_]                          # the line's free cameras (0: constant line / none)
            if not mask:
                continue
            cams = [c for c in range(10) if (mask >> c) & 1]
            assert a == cams[0] and wdt == cams[-1] - cams[0] + 1 and nb == (6 * wdt + 15) // 16 and holes == int(len(cams) != wdt)
            if a != cur_a:
                if cur_a >= 0:
                    flush(acc, RC6, cur_a); nflush += 1
                    if row3_used:
                        flush(row3, [(3, c) for c in range(4)], cur_a); row3_used = False
                cur_a = a
                groups_seen.append(a)
            X = np.zeros((4, 64))
            for r in range(nb):
                for lane in range(64):
                    X[r, lane] = panel[host_math.hm_gp_fetch(lane, r, d)]
            assert not np.isnan(X).any()
            stack = np.zeros((64, 4))
            for c in cams:
                stack[6 * c:6 * c + 6] = F[(s_, c)]
            want += stack @ stack.T
            for e, (r, c) in enumerate(RC6):
                if r < nb:
                    acc[e] += product(X[r], X[c]); nmfma += 1
            if nb >= 4:
                for c in range(4):
                    row3[c] += product(X[3], X[c]); nmfma += 1
                row3_used = True
        assert len(seen_lines) == nl
        if row3_used:
            flush(row3, [(3, c) for c in range(4)], cur_a)
    if cur_a >= 0:
        flush(acc, RC6, cur_a)
    got = np.zeros((64, 64))
    row, col = C.c_int(0), C.c_int(0)
    for tt in range(10):
        for qq in range(4):
            for lane in range(64):
                host_math.hm_acc_rc(tt, qq, lane, C.byref(row), C.byref(col))
                got[row.value, col.value] = slab[tt * 256 + qq * 64 + lane]
    lower = np.tril(np.ones((64, 64), dtype=bool))
    assert abs(got - want)[lower].max() < 1e-11 and abs(want[n:, :]).max() == 0.0
    if which == "sliding":
        assert groups_seen == sorted(groups_seen) or nflush <= 2 * ncf      # the groups follow each other: a handful of adds to memory per window
        assert nflush <= 3 * ncf and 2.5 < nmfma / L < 5.0


def test_raw_linearisation_matches_the_standard_one(host_math, oracle):
    """obs_linearise_raw (the grouped sweep: raw camera coordinates, line frame in camera coordinates, Huber factor folded into the
    row gradients, line scale folded into the line's columns) against obs_linearise + the explicit products: r sqrt(rho'),
    J_c' JL = J_c sqrt(rho') (dr/dw = tau^T JL(w)), J_l diag(sl) sqrt(rho'); inliers, outliers, loss off, the identity keyframe."""
    rng = np.random.default_rng(3)
    a = 1.0 / 406.05
    for case in range(40):
        cam = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 1.0, 3)])
        if case % 5 == 0: cam[:3] = 0.0
        line = np.array([rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3), rng.uniform(0.2, 1.3)])
        obs = rng.normal(0, 0.3 if case % 2 else 0.002, 8)
        sl = rng.uniform(0.1, 1.0, 4)
        delta = 0.0 if case % 7 == 0 else a
        r, jc, jl = np.zeros(4), np.zeros(24), np.zeros(16)
        host_math.hm_obs_linearise(_dp(cam), _dp(line), _dp(obs), C.c_double(0.12), _dp(r), _dp(jc), _dp(jl))
        R, JL = np.zeros(9), np.zeros(9)
        host_math.hm_cam_prepare(_dp(cam), _dp(R), _dp(JL))
        JL = JL.reshape(3, 3)
        cost_std = C.c_double(0)
        host_math.hm_huber.restype = C.c_double
        sr = host_math.hm_huber(C.c_double(float(r @ r)), C.c_double(delta), C.byref(cost_std))
        rs, jcr, jlr, cost = np.zeros(4), np.zeros(24), np.zeros(16), C.c_double(0)
        host_math.hm_obs_linearise_raw(_dp(cam), _dp(line), _dp(obs), C.c_double(0.12), C.c_double(delta), _dp(sl), _dp(rs), _dp(jcr), _dp(jlr), C.byref(cost))
        jc, jl, jcr, jlr = jc.reshape(4, 6), jl.reshape(4, 4), jcr.reshape(4, 6), jlr.reshape(4, 4)
        scale = max(1.0, np.abs(jc).max(), np.abs(jl).max())
        assert np.abs(rs - sr * r).max() < 1e-13 * max(1.0, np.abs(r).max())
        assert abs(cost.value - cost_std.value) <= 1e-14 * max(cost_std.value, 1e-30)
        assert np.abs(jcr[:, :3] @ JL - sr * jc[:, :3]).max() < 1e-11 * scale
        assert np.abs(jcr[:, 3:] - sr * jc[:, 3:]).max() < 1e-11 * scale
        assert np.abs(jlr - sr * jl * sl[None, :]).max() < 1e-11 * scale


def test_mixed_precision_linearisation(host_math):
    """obs_linearise_raw_mixed (lba_precision = 1): residuals, Huber factor, block cost and the LINE Jacobian are the double routine's to
    round-off - also for far lines (t -> 0: d = cos t / sin t of reference src/lba_problem.h:63 in the hundreds) and lines that pass close
    to the principal point (the normalisation of :90); the CAMERA Jacobian, formed in float, agrees with the double one to a few float
    ulps of the row's largest entry (stated: 2e-6), far lines included - its rotation part is formed as tau = q x (dc x Q) + dc x (q x t_k),
    without the cancellation of Q x gP + dc x gD.  What the float J_c' does to a solve: tests/test_gpu_lba.py::test_mixed_precision_solves."""
    rng = np.random.default_rng(11)
    a = 1.0 / 406.05
    worst = {"near": 0.0, "far": 0.0}
    for case in range(160):
        cam = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 1.0, 3)])
        if case % 5 == 0: cam[:3] = 0.0
        far = case % 3 == 0
        t = rng.uniform(0.002, 0.02) if far else rng.uniform(0.2, 1.3)          # every third line is far away (d = 50 .. 500)
        line = np.array([rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3), t])
        obs = rng.normal(0, 0.3 if case % 2 else 0.002, 8)
        sl = rng.uniform(0.1, 1.0, 4)
        delta = 0.0 if case % 7 == 0 else a
        rs, jc, jl, cost = np.zeros(4), np.zeros(24), np.zeros(16), C.c_double(0)
        rm, jcm, jlm, costm = np.zeros(4), np.zeros(24), np.zeros(16), C.c_double(0)
        host_math.hm_obs_linearise_raw(_dp(cam), _dp(line), _dp(obs), C.c_double(0.12), C.c_double(delta), _dp(sl), _dp(rs), _dp(jc), _dp(jl), C.byref(cost))
        host_math.hm_obs_linearise_raw_mixed(_dp(cam), _dp(line), _dp(obs), C.c_double(0.12), C.c_double(delta), _dp(sl), _dp(rm), _dp(jcm), _dp(jlm), C.byref(costm))
        assert np.abs(rs - rm).max() <= 1e-15 * max(1.0, np.abs(rs).max())
        assert abs(cost.value - costm.value) <= 1e-15 * max(cost.value, 1e-300)
        assert np.abs(jl - jlm).max() <= 1e-13 * max(1.0, np.abs(jl).max())
        f, m = jc.reshape(4, 6), jcm.reshape(4, 6)
        for row in range(4):
            dev = np.abs(f[row] - m[row]).max() / (np.abs(f[row]).max() + 1e-300)
            worst["far" if far else "near"] = max(worst["far" if far else "near"], dev)
    assert worst["near"] < 2e-6 and worst["far"] < 2e-6, worst


def test_backsub_contraction_matches_jacobians(host_math):
    """The back-substitution's w = J_l^T (J_c y_c), contracted on the fly (obs_backsub_w), equals the product of the
    explicit analytic Jacobians."""
    rng = np.random.default_rng(3)
    w = synth.make_window(5, num_lines=40)
    Cn, prm = w["num_cameras"], w["parameters"]
    for i in range(0, len(w["camera_index"]), 3):
        cam = prm[6 * w["camera_index"][i]:][:6].copy()
        line = prm[6 * Cn + 4 * w["line_index"][i]:][:4].copy()
        ob = w["observations"][i].copy()
        y = rng.normal(size=6)
        r, jc, jl, r2, wv = np.zeros(4), np.zeros((4, 6)), np.zeros((4, 4)), np.zeros(4), np.zeros(4)
        host_math.hm_obs_linearise(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(r), _dp(jc), _dp(jl))
        host_math.hm_obs_backsub_w(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(y), _dp(r2), _dp(wv))
        want = jl.T @ (jc @ y)
        assert abs(r - r2).max() < 1e-15 and abs(wv - want).max() < 1e-12 * (1 + abs(want).max())


@pytest.mark.parametrize("scale", [0.0, 1e-12, 1e-3, 0.5, 3.0])
def test_rotation_branches(host_math, oracle, scale):
    """theta == 0 (the identity keyframe, slam.cpp:1322) takes the first-order branch of
    AngleAxisRotatePoint; tiny and large angles go through Rodrigues."""
    rng = np.random.default_rng(1)
    w = synth.make_window(4, num_lines=20)
    line = w["parameters"][6 * 20:6 * 20 + 4].copy()
    ob = w["observations"][0].copy()
    cam = np.concatenate([rng.normal(size=3) * scale, rng.normal(size=3)])
    r0, jc0, jl0 = oracle.line_residual_jet(cam, line, ob)
    r, jc, jl = np.zeros(4), np.zeros((4, 6)), np.zeros((4, 4))
    host_math.hm_obs_linearise(_dp(cam), _dp(line), _dp(ob), C.c_double(0.12), _dp(r), _dp(jc), _dp(jl))
    assert abs(r - r0).max() < 1e-14
    assert abs(jc - jc0).max() < 1e-12 * (1 + abs(jc0).max())
    assert abs(jl - jl0).max() < 1e-12 * (1 + abs(jl0).max())


def test_huber_corrector(host_math, oracle):
    a = 1.0 / 406.05
    for s in (0.0, 0.3 * a * a, a * a, 4 * a * a, 1.0):
        cost = C.c_double(0)
        host_math.hm_huber.restype = C.c_double
        f = host_math.hm_huber(C.c_double(s), C.c_double(a), C.byref(cost))
        rho = oracle.huber(s, a)
        assert abs(f - np.sqrt(rho[1])) < 1e-15 and abs(cost.value - 0.5 * rho[0]) < 1e-18
    f = host_math.hm_huber(C.c_double(1.0), C.c_double(0.0), C.byref(cost))   # FLAGS_robust = false
    assert f == 1.0 and cost.value == 0.5


def _pack(host_math, w, grouping=0):
    Cn, L = int(w["num_cameras"]), int(w["num_lines"])
    cam = np.ascontiguousarray(w["camera_index"], dtype=np.int32)
    line = np.ascontiguousarray(w["line_index"], dtype=np.int32)
    fixed = np.ascontiguousarray(w["fixed_index"], dtype=np.int32)
    obs = np.ascontiguousarray(w["observations"], dtype=np.float64).reshape(-1)
    prm = np.ascontiguousarray(w["parameters"], dtype=np.float64).copy()
    M = len(cam)
    counts = np.zeros(5, dtype=np.int32)
    lo, lp = np.zeros(max(L, 1), dtype=np.int32), np.zeros(L + 1, dtype=np.int32)
    oo, oc, cf = np.zeros(max(M, 1), dtype=np.int32), np.zeros(max(M, 1), dtype=np.int32), np.zeros(max(Cn, 1), dtype=np.int32)
    max_tiles, max_items = L + 8, 64 * M + 8
    tiles = np.zeros(4 * max_tiles, dtype=np.int32)
    items = np.zeros(2 * max_items, dtype=np.uint8)
    lane_map = np.zeros(64 * max_tiles, dtype=np.uint16)
    desc = np.zeros(max(L, 1), dtype=np.uint32)
    rc = host_math.hm_pack_g(Cn, L, M, _ip(cam), _ip(line), _ip(fixed), _dp(obs), _dp(prm), _ip(counts), _ip(lo), _ip(lp),
                             _ip(oo), _ip(oc), _ip(tiles), items.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cf), max_tiles, max_items,
                             lane_map.ctypes.data_as(C.POINTER(C.c_ushort)), grouping, desc.ctypes.data_as(C.POINTER(C.c_uint)))
    return rc, dict(desc=desc[:L], Cf=int(counts[0]), ntiles=int(counts[1]), nitems=int(counts[2]), nfree=int(counts[3]), nkept=int(counts[4]),
                    line_order=lo[:L], line_ptr=lp, ob_orig=oo[:M], ob_cam=oc[:M], cam_cf=cf[:Cn],
                    tiles=tiles[:4 * int(counts[1])].reshape(-1, 4), items=items[:2 * int(counts[2])].reshape(-1, 2),
                    lane_map=lane_map[:64 * int(counts[1])].reshape(-1, 64))


def test_pack_invariants(host_math):
    """LBAProblem::build replacement (lba_pack.cpp): a permutation of lines/observations into
    64-lane tiles plus the camera-pair work items; nothing lost, nothing duplicated."""
    w = synth.make_window(6, num_lines=300)
    rc, P = _pack(host_math, w)
    assert rc == 0
    L, M = 300, len(w["camera_index"])
    assert P["Cf"] == 10 and P["nkept"] == M and P["nfree"] == 60 + 4 * L
    assert sorted(P["line_order"]) == list(range(L)) and sorted(P["ob_orig"]) == list(range(M))
    assert P["line_ptr"][0] == 0 and P["line_ptr"][L] == M
    for s in range(L):
        sel = P["ob_orig"][P["line_ptr"][s]:P["line_ptr"][s + 1]]
        assert np.all(w["line_index"][sel] == P["line_order"][s])        # grouped by line
        cf = P["cam_cf"][w["camera_index"][sel]]
        nf = int((cf >= 0).sum())
        assert np.all(cf[:nf] >= 0) and np.all(cf[nf:] < 0) and np.all(np.diff(cf[:nf]) >= 0)   # free first, ascending
    assert np.array_equal(P["ob_cam"], w["camera_index"][P["ob_orig"]])
    _check_tiles(P, w, L)


def _check_tiles(P, w, L, with_items=True):
    """Tiles cover the sorted lines exactly once; every line owns a run of max(k, 1) consecutive lanes that stays inside
    one 16-lane row unless it starts on a row boundary; the pair items name the lanes of the run's free cameras."""
    covered, it = 0, 0
    for t, (lb, nl, flags, ni) in enumerate(P["tiles"]):
        assert lb == covered and 1 <= nl <= 64
        m = P["lane_map"][t]
        slot, pos = m & 0xFF, (m >> 8) & 0x3F
        want, min_run, max_run, multi = set(), 64, 1, 0
        for q in range(nl):
            s = lb + q
            k = int(P["line_ptr"][s + 1] - P["line_ptr"][s])
            run = max(k, 1)
            lanes = np.nonzero(slot == q)[0]
            assert len(lanes) == run and np.array_equal(lanes, lanes[0] + np.arange(run))
            assert np.array_equal(pos[lanes], np.arange(run))
            assert lanes[0] // 16 == lanes[-1] // 16 or lanes[0] % 16 == 0
            if q: assert lanes[0] > np.nonzero(slot == q - 1)[0][-1]          # runs in line order
            min_run, max_run, multi = min(min_run, run), max(max_run, min(run, 16)), multi or run > 16
            sel = P["ob_orig"][P["line_ptr"][s]:P["line_ptr"][s + 1]]
            kf = int((P["cam_cf"][w["camera_index"][sel]] >= 0).sum())
            if not np.any(np.asarray(w["fixed_index"]).reshape(-1, 2)[sel, 1]):
                want |= {(lanes[0] + i, lanes[0] + j) for i in range(kf) for j in range(i + 1, kf)}
        assert np.all(slot[slot != 0xFF] < nl)
        # skew flag (bit 15): inside a 16-lane row the lanes that hold observations of one free camera alternate 0, 1, 0, ...
        skew = (m >> 15) & 1
        for row in range(4):
            seen = {}
            for lane in range(16 * row, 16 * row + 16):
                if slot[lane] == 0xFF:
                    assert skew[lane] == 0
                    continue
                sl = lb + int(slot[lane])
                kk = int(P["line_ptr"][sl + 1] - P["line_ptr"][sl])
                if pos[lane] >= kk:
                    assert skew[lane] == 0
                    continue
                cf = int(P["cam_cf"][P["ob_cam"][P["line_ptr"][sl] + int(pos[lane])]])
                if cf < 0:
                    assert skew[lane] == 0
                    continue
                assert skew[lane] == seen.get(cf, 0) % 2
                seen[cf] = seen.get(cf, 0) + 1
        assert (flags & 1) == int(multi) and (flags >> 3) & 31 == max_run
        assert 1 << ((flags >> 1) & 3) == (1 if min_run >= 4 else 2 if min_run >= 2 else 4)
        got = [tuple(int(v) for v in x) for x in P["items"][it:it + ni]]
        assert len(got) == len(set(got)) and set(got) == (want if with_items else set())
        it += ni
        covered += nl
    assert covered == L and it == P["nitems"]


def test_pack_long_and_short_lines(host_math):
    """Lines with more than 16 observations take whole rows; lines with fewer than 4 keep to tiles of their own."""
    rng = np.random.default_rng(5)
    Cn, L = 64, 60
    counts = np.concatenate([[64, 49, 48, 33, 32, 17], rng.integers(4, 17, 28), [16, 16], rng.integers(0, 4, 24)])
    cam, line = [], []
    for l, k in enumerate(counts):
        cam += list(rng.choice(Cn, size=k, replace=False)); line += [l] * k
    M = len(cam)
    fixed = np.zeros((M, 2), dtype=np.int32); fixed[np.asarray(cam) >= 12, 0] = 1
    w = dict(num_cameras=Cn, num_lines=L, camera_index=np.asarray(cam, dtype=np.int32), line_index=np.asarray(line, dtype=np.int32),
             fixed_index=fixed.reshape(-1), observations=rng.normal(size=(M, 8)), parameters=rng.normal(size=6 * Cn + 4 * L))
    rc, P = _pack(host_math, w)
    assert rc == 0 and P["Cf"] == 12
    _check_tiles(P, w, L)
    # a line with more than 64 observations does not fit a wave: the window goes to the global-memory path
    # (lba_big.h) - no tiles, no pair items, lines in their original order, observations still grouped by line
    too_long = dict(w, camera_index=np.concatenate([w["camera_index"], (np.arange(65) % 64).astype(np.int32)]),
                    line_index=np.concatenate([w["line_index"], np.full(65, 59, dtype=np.int32)]),
                    fixed_index=np.concatenate([w["fixed_index"], np.stack([np.arange(65) % 64 >= 12, np.zeros(65)], 1).astype(np.int32).reshape(-1)]),
                    observations=np.concatenate([w["observations"], rng.normal(size=(65, 8))]),
                    parameters=w["parameters"])
    rc2, P2 = _pack(host_math, too_long)
    assert rc2 == 0 and P2["ntiles"] == 0 and P2["nitems"] == 0 and P2["Cf"] == 12
    assert np.array_equal(P2["line_order"], np.arange(L))
    assert np.array_equal(np.diff(P2["line_ptr"]), np.bincount(too_long["line_index"], minlength=L))
    assert sorted(P2["ob_orig"].tolist()) == list(range(len(too_long["camera_index"])))
    runs = {}
    for t, (lb, nl, flags, ni) in enumerate(P["tiles"]):
        ks = [max(int(P["line_ptr"][s + 1] - P["line_ptr"][s]), 1) for s in range(lb, lb + nl)]
        assert max(ks) < 4 or min(ks) >= 4                     # short lines never share a tile with the others


@pytest.mark.parametrize("nfree", [21, 40, 45])
def test_pack_oversize_windows(host_math, nfree):
    """Windows beyond the tiled sweeps (more than 20 free cameras: the reference's W = 40 study) go through the same packer
    (validation, constness, free indices, observations grouped by line with the free cameras first) and come out without tiles
    or pair items: the global-memory path builds its own gather lists.  (Also run under ASan / UBSan: tests/test_sanitizers.py.)"""
    w = synth.make_window(40 + nfree, num_lines=50, num_kf=2 * nfree, num_free=nfree, mean_track=30.0)
    rc, P = _pack(host_math, w)
    L, M = int(w["num_lines"]), len(w["camera_index"])
    assert rc == 0 and P["Cf"] == nfree and P["ntiles"] == 0 and P["nitems"] == 0
    assert np.array_equal(P["line_order"], np.arange(L)) and sorted(P["ob_orig"].tolist()) == list(range(M))
    assert P["nfree"] == 6 * nfree + 4 * L and P["nkept"] == M
    for s_ in range(L):
        sel = P["ob_orig"][P["line_ptr"][s_]:P["line_ptr"][s_ + 1]]
        assert np.all(np.asarray(w["line_index"])[sel] == s_)
        cf = P["cam_cf"][np.asarray(w["camera_index"])[sel]]
        nf = int((cf >= 0).sum())
        assert np.all(cf[:nf] >= 0) and np.all(cf[nf:] < 0) and np.all(np.diff(cf[:nf]) >= 0)
    rc2, P2 = _pack(host_math, w, grouping=1)               # asking for the grouped order changes nothing for such a window
    assert rc2 == 0 and P2["ntiles"] == 0 and np.array_equal(P2["line_order"], P["line_order"]) and np.array_equal(P2["ob_orig"], P["ob_orig"])


def test_pack_edge_cases(host_math):
    # motion-only shape: lines constant -> no elimination work items, 1 free camera
    rc, P = _pack(host_math, synth.make_motion_only(1, num_lines=25))
    assert rc == 0 and P["Cf"] == 1 and P["nitems"] == 0 and P["nfree"] == 6
    assert P["nkept"] == 25                      # the (fixed camera, fixed line) blocks leave the program
    # empty window
    rc, P = _pack(host_math, dict(num_cameras=2, num_lines=3, camera_index=[], line_index=[], fixed_index=[],
                                  observations=np.zeros((0, 8)), parameters=np.zeros(24)))
    assert rc == 0 and P["ntiles"] >= 1 and P["nfree"] == 0
    # out-of-range index and non-finite input are rejected
    w = synth.make_window(2, num_lines=10)
    bad = dict(w, camera_index=w["camera_index"].copy()); bad["camera_index"][0] = 99
    assert _pack(host_math, bad)[0] == 1
    bad = dict(w, observations=w["observations"].copy()); bad["observations"][0, 0] = np.nan
    assert _pack(host_math, bad)[0] == 1
    # more free cameras than the LDS-resident reduced system holds: packed for the global-memory path, no tiles
    big = synth.make_window(2, num_lines=30, num_kf=24, num_free=24)
    rc, P = _pack(host_math, big)
    assert rc == 0 and P["Cf"] == 24 and P["ntiles"] == 0 and P["nitems"] == 0 and P["nfree"] == 6 * 24 + 4 * 30


def _declared_functions(header="slslam_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slslam_[a-z_0-9]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    from slslam_amd import capi
    L = capi.lib()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "include/slslam_hip.h declares %s but the library does not export it" % n
    assert sorted(capi.EXPORTS) == names
    assert b"gfx950" in L.slslam_version()


def test_dist_library_exports_every_declared_symbol():
    """include/slslam_dist.h (the C-level multi-GPU fan-out: host C++ on libslslam_hip.so + librccl): the library builds, loads and
    exports what the header declares; without a device its entry points report it (no compute call is made here)."""
    import ctypes as C
    import subprocess
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "slslam_amd", "dist_c")])
    L = C.CDLL(os.path.join(ROOT, "slslam_amd", "_lib", "libslslam_dist.so"))
    names = _declared_functions("slslam_dist.h")
    assert names == ["slslam_dist_create", "slslam_dist_debug_fail_next_shard", "slslam_dist_destroy", "slslam_dist_rank", "slslam_dist_shard_range",
                     "slslam_dist_solve", "slslam_dist_stream_collect", "slslam_dist_stream_create", "slslam_dist_stream_destroy", "slslam_dist_stream_submit",
                     "slslam_dist_unique_id", "slslam_dist_world"]
    for n in names:
        assert hasattr(L, n), "include/slslam_dist.h declares %s but libslslam_dist.so does not export it" % n
    L.slslam_dist_shard_range.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.slslam_dist_shard_range.restype = None
    from slslam_amd.dist import shard_range
    for n, world in ((8192, 8), (10, 4), (3, 8), (0, 2)):
        for r in range(world):
            lo, hi = C.c_longlong(), C.c_longlong()
            L.slslam_dist_shard_range(n, r, world, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == tuple(shard_range(n, r, world))
    from slslam_amd import capi
    if capi.device_count() == 0:
        out = C.c_void_p()
        ident = (C.c_ubyte * 128)()
        assert L.slslam_dist_create(0, 1, 0, ident, C.byref(out)) == 2          # SLSLAM_ERR_NO_DEVICE


def test_default_options_are_the_reference_configuration():
    from slslam_amd import capi
    o = capi.default_options()
    assert o.max_num_iterations == 10                         # main.cpp:23
    assert abs(o.huber_delta - 1.0 / 406.05) < 1e-18          # lba_problem.cpp:78
    assert o.baseline == 0.12                                 # lba_problem.h:101
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.function_tolerance) == (1e4, 1e-3, 1e-6)
    with pytest.raises(TypeError):
        capi.default_options(no_such_option=1)


def test_no_cpu_fallback():
    """Without a HIP device the product must fail loudly, never route through a CPU path."""
    from slslam_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    w = synth.make_window(0, num_lines=10)
    with pytest.raises(capi.SlslamError) as e:
        capi.lba_solve(w)
    assert e.value.status == 2
    with pytest.raises(capi.SlslamError):
        capi.LBABatch()
    # the product never imports, links or calls the oracle
    pat = re.compile(r"(import\s+oracle|from\s+oracle|pyoracle|liboracle|slslam_oracle\.h|oracle_[a-z_]+\()")
    for d in ("slslam_amd", os.path.join("slslam_amd", "csrc"), os.path.join("slslam_amd", "host"), "include"):
        p = os.path.join(ROOT, d)
        if not os.path.isdir(p):
            continue
        for f in os.listdir(p):
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                assert not pat.search(open(os.path.join(p, f)).read()), "%s/%s references the oracle" % (d, f)


def test_argument_validation_comes_before_the_device():
    """Malformed inputs are rejected with SLSLAM_ERR_INVALID_ARGUMENT (1) on any machine; well-formed ones need a
    HIP device (SLSLAM_ERR_NO_DEVICE = 2 here): every entry point, including the RANSAC and pose-graph ones."""
    from slslam_amd import capi
    gpu = capi.device_count() > 0

    def status(fn, *a, **k):
        try:
            fn(*a, **k)
            return 0
        except capi.SlslamError as e:
            return e.status
    w = synth.make_window(1, num_lines=8)
    bad = dict(w, camera_index=w["camera_index"].copy())
    bad["camera_index"][0] = 99
    assert status(capi.lba_solve, bad) == 1
    nanw = dict(w, observations=w["observations"].copy())
    nanw["observations"][0, 0] = np.nan
    assert status(capi.lba_solve, nanw) == 1
    assert status(capi.lba_solve, w, max_num_iterations=-1) == 1
    g = synth.make_pose_graph(1, num_poses=6, num_loops=1)
    gb = dict(g, pose_index_2=g["pose_index_2"].copy())
    gb["pose_index_2"][0] = gb["pose_index_1"][0]                         # self edge
    assert status(capi.po_solve, gb) == 1
    fr = synth.make_ransac_pair(1, num_lines=20, num_trials=4)
    smp = fr["samples"].copy()
    smp[0, 0] = 20                                                        # index past the common lines
    assert status(capi.ransac_generate, fr["obs0"], fr["obs1"], smp) == 1
    assert status(capi.ransac_motion, fr["obs0"], fr["obs1"], fr["lines"], smp) == 1
    assert status(capi.ransac_generate, fr["obs0"], fr["obs1"], np.zeros((2, 17), dtype=np.int32)) == 1    # s > 16
    ok = 0 if gpu else 2
    assert status(capi.lba_solve, w) == ok
    assert status(capi.po_solve, g) == ok
    assert status(capi.ransac_generate, fr["obs0"], fr["obs1"], fr["samples"]) == ok
    assert status(capi.ransac_motion, fr["obs0"], fr["obs1"], fr["lines"], fr["samples"]) == ok
    poses, obs, lines, _ = synth.make_ransac_frame(1, num_lines=10, num_hypotheses=3)
    assert status(capi.ransac_score, poses, obs, lines) == ok


def _check_po_structure(g, st):
    """invariants of the chains-first ordering (slslam_po_structure)"""
    N = int(g["num_poses"])
    gauge = int(g["pose_index_1"][0])
    used = np.zeros(N, bool)
    used[g["pose_index_1"]] = True; used[g["pose_index_2"]] = True
    free = used.copy(); free[gauge] = False
    slot = st["slot"]
    assert np.array_equal(slot >= 0, free)
    assert sorted(slot[free]) == list(range(0, 6 * int(free.sum()), 6)) and st["num_unknowns"] == 6 * int(free.sum())
    pose_of = {int(s): k for k, s in enumerate(slot) if s >= 0}
    adj = {k: set() for k in range(N) if free[k]}
    for a, b in zip(g["pose_index_1"], g["pose_index_2"]):
        a, b = int(a), int(b)
        if free[a] and free[b]:
            adj[a].add(b); adj[b].add(a)
    in_chain = {}
    covered = 0
    n1 = st["level1_chains"]
    for ci, (start, ln, left, right) in enumerate(st["chains"]):
        assert 1 <= ln <= 32 and start == covered                      # chains are laid out back to back, level after level; at most 32 poses each
        covered += 6 * ln
        poses = [pose_of[start + 6 * i] for i in range(ln)]
        for j in (left, right):
            assert j < 0 or j >= start + 6 * ln                         # what a chain ends at comes later in the ordering (a cut pose of a higher level or a junction)
        assert left < 0 or left != right
        for i, v in enumerate(poses):
            assert v not in in_chain
            in_chain[v] = ci
            if ci >= n1:
                continue                                                # a chain of cut poses: consecutive poses are joined by a piece, not by an edge
            nb = set(adj[v])
            if i > 0:
                assert poses[i - 1] in nb; nb.discard(poses[i - 1])     # consecutive poses of a level-1 chain are connected
            if i + 1 < ln:
                assert poses[i + 1] in nb; nb.discard(poses[i + 1])
            ends = set()
            if i == 0 and left >= 0:
                ends.add(pose_of[left])
            if i == ln - 1 and right >= 0:
                ends.add(pose_of[right])
            assert nb == ends, (v, nb, ends)                            # every other neighbour is what the chain ends at on that side
    assert covered == st["num_chain_unknowns"]
    # the poses of the higher-level chains are cut poses: each is the end of a chain of a lower level, and has exactly two graph neighbours
    ends_of_lower = {}
    for ci, (start, ln, left, right) in enumerate(st["chains"]):
        for j in (left, right):
            if 0 <= j < st["num_chain_unknowns"]:
                ends_of_lower.setdefault(pose_of[j], []).append(ci)
    for ci, (start, ln, left, right) in enumerate(st["chains"][n1:], start=n1):
        for i in range(ln):
            v = pose_of[start + 6 * i]
            assert len(adj[v]) == 2 and v in ends_of_lower and all(c < ci for c in ends_of_lower[v]), (v, ci)
    for k in range(N):                                                  # what is not on a chain is a junction
        if free[k] and k not in in_chain:
            assert slot[k] >= st["num_chain_unknowns"]


def test_po_structure_topologies():
    """Host-side symbolic analysis of the structured pose-graph factorisation, no device needed."""
    from slslam_amd import capi
    rng = np.random.default_rng(3)

    def graph(n, extra):
        p1 = list(range(n - 1)) + [min(a, b) for a, b in extra]
        p2 = list(range(1, n)) + [max(a, b) for a, b in extra]
        order = sorted(range(len(p1)), key=lambda i: (p1[i], p2[i]))
        return {"num_poses": n, "pose_index_1": np.array([p1[i] for i in order], dtype=np.int32),
                "pose_index_2": np.array([p2[i] for i in order], dtype=np.int32)}
    cases = [graph(50, []), graph(30, [(1, 29)]), graph(30, [(5, 14), (5, 20)]), graph(40, [(10, k) for k in (15, 20, 25, 30, 35, 39)]),
             graph(120, [(3, 110), (7, 100)]), graph(60, [(i, i + 7) for i in range(1, 50, 3)]), graph(2, []), graph(300, [])]
    for _ in range(20):
        n = int(rng.integers(3, 200))
        extra = [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(int(rng.integers(0, 12)))]
        cases.append(graph(n, [e for e in extra if e[1] - e[0] > 1]))
    for g in cases:
        _check_po_structure(g, capi.po_structure(g))
    # a bare path of 299 free poses: three levels of chains (pieces of ~299^(1/3) poses, the chains of their cut poses), NO junction block, and
    # a sequential depth - the longest chain of every level, added up - of about 3 x 299^(1/3) steps (until round 5: ten chains of 32 + 9 junction poses)
    st = capi.po_structure(graph(300, []))
    assert st["num_unknowns"] == st["num_chain_unknowns"] == 6 * 299
    n1 = st["level1_chains"]
    lens1 = [c[1] for c in st["chains"][:n1]]
    assert max(lens1) <= 8 and n1 >= 30
    upper = [c[1] for c in st["chains"][n1:]]
    assert upper and sum(upper) == 299 - sum(lens1) and max(upper) <= 8
    # the bench graph: its 8 loop closures leave 10 junction poses - one 64-wide block of the dense factorisation
    st = capi.po_structure(synth.make_pose_graph(7, num_poses=260, num_loops=8))
    assert st["num_unknowns"] - st["num_chain_unknowns"] == 60
    # malformed graphs
    bad = graph(5, [])
    bad["pose_index_2"] = bad["pose_index_2"].copy(); bad["pose_index_2"][1] = 7
    with pytest.raises(capi.SlslamError) as e:
        capi.po_structure(bad)
    assert e.value.status == 1


def test_sweep_kernels_keep_their_occupancy(tmp_path):
    """The two observation sweeps sit exactly at the register budget of 2 waves per SIMD (256 VGPRs, fp64 values cost two):
    a few more live values anywhere in them halve the occupancy or spill, which costs 20-40 % of the bench figure without
    failing any numerical test.  Compile the device code the way the build does and read the resource summary the compiler
    prints for every kernel."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "lba_api.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
                           "-o", out, os.path.join(ROOT, "slslam_amd", "csrc", "lba_api.hip")], stderr=subprocess.DEVNULL)
    stats, name = {}, None
    for line in open(out):
        m = re.match(r"^(_ZN6slslam\S+):", line)
        if m:
            name = m.group(1)
        m = re.match(r"^; (TotalNumVgprs|ScratchSize|Occupancy): (\d+)", line)
        if m and name:
            stats.setdefault(name, {})[m.group(1)] = int(m.group(2))
    def find(fragment):
        hits = [v for k, v in stats.items() if fragment in k]
        assert len(hits) == 1, (fragment, [k for k in stats if fragment in k])
        return hits[0]
    # the elimination sweep's three forms: first sweep of a solve (FRESH = 1), steady (0), run-time flag (-1)
    for frag in ("17k_linearise_schurILb0ELi1E", "17k_linearise_schurILb0ELi0E", "17k_linearise_schurILb0ELin1E", "9k_backsubE"):
        st = find(frag)
        assert st["Occupancy"] == 2 and st["ScratchSize"] == 0 and st["TotalNumVgprs"] <= 256, (frag, st)
    assert find("15k_reduced_solveILi4E")["Occupancy"] >= 4      # four workgroups per CU: the whole bench batch resident in one round
    assert find("15k_reduced_solveILi1E")["ScratchSize"] == 0    # the one-window-per-CU form: the tile factorisation stays in registers


def test_graded_chunk_boundaries(host_math):
    """chunk_boundaries_graded (lba_pack.cpp; what finalize cuts the windows of a batch with when every wave slot runs several chunks):
    the tile counts follow the weights, every chunk has at least one tile, the cut covers the window exactly, it is a function of
    (tiles, chunks, weights) alone - a window solved again alone gets the same chunks - and degenerate requests fall back to the equal cut."""
    def graded(nt, w):
        out = (C.c_int * 1100)()
        n = host_math.hm_chunk_boundaries_graded(nt, len(w), (C.c_int * len(w))(*w), out, 1100)
        return list(out[:n])
    def equal(nt, per):
        out = (C.c_int * 1100)()
        n = host_math.hm_chunk_boundaries(nt, per, out, 1100)
        return list(out[:n])
    b = graded(200, [84, 84, 24, 24, 12, 12])                      # the bench window: three rounds of the slots, six chunks
    assert b == [0, 70, 140, 160, 180, 190, 200]
    b = graded(200, [84, 84, 84, 84, 36, 36, 36, 36])              # two rounds, eight chunks: 35 / 15 tiles
    assert [y - x for x, y in zip(b, b[1:])] == [35, 35, 35, 35, 15, 15, 15, 15]
    rng = np.random.default_rng(5)
    for _ in range(300):
        nc = int(rng.integers(2, 40)); nt = int(rng.integers(1, 1500))
        w = [int(x) for x in rng.integers(1, 100, size=nc)]
        b = graded(nt, w)
        assert b[0] == 0 and b[-1] == nt and all(y > x for x, y in zip(b, b[1:])), (nt, w, b)
        assert b == graded(nt, w)
        if nt >= 2 * nc:
            assert len(b) == nc + 1
            sizes = np.diff(b); ideal = nt * np.array(w) / sum(w)
            assert np.all(np.abs(sizes - ideal) <= np.maximum(2.0, 0.0)) or np.all(sizes >= 1)       # rounding of the running sums: within two tiles unless clamped to one
        else:
            assert b == equal(nt, (nt + nc - 1) // nc)
    assert graded(0, [1, 1]) == [0]
    #  the fallback of a graded request on a small window is reported as a POSITIVE chunk count m (slslam_lba_batch_window_chunks);
    # handed back as chunks_per_window = m it must cut the window the same way: per' = ceil(nt / m) with m = ceil(nt / per), per = ceil(nt / nc)
    for nt in range(1, 260):
        for nc in range(2, 41):
            if nt >= 2 * nc:
                continue
            b = graded(nt, [1] * nc)
            m = len(b) - 1
            assert b == equal(nt, (nt + m - 1) // m), (nt, nc, b)


def test_packer_layout_matches_the_golden_digest(host_math):
    """Everything pack_window emits for a fixed family of windows (bench shapes, wide tracks, 20 free cameras, motion-only, scrambled order
    with holes and constant lines), both packings, against tests/golden/packer_digest.json (made by tests/golden/make_packer_digest.py): a
    window's solved bytes are a function of its packed layout, so host-side work on the packer (round 5: the same layout 37 % faster) must
    not move a byte of it."""
    import importlib.util
    import json
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_packer_digest", os.path.join(gold_dir, "make_packer_digest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold = json.load(open(os.path.join(gold_dir, "packer_digest.json")))
    seen = 0
    for name, w in mod.family().items():
        for g in (0, 1):
            rc, P = _pack(host_math, w, grouping=g)
            assert rc == 0
            ref = gold["%s/grouping%d" % (name, g)]
            assert (P["ntiles"], P["nitems"]) == (ref["tiles"], ref["items"]), (name, g)
            assert mod.digest(P) == ref["crc32"], (name, g)
            seen += 1
    assert seen == len(gold) == 12


def test_pack_observation_destinations(host_math):
    """Where a pack's observations go (lba_pack.h::ObPlanes): into the packed window, into caller planes (a refill that gathers on the host), or
    - a refill whose batch permutes on the device (lba_api.hip::k_permute_obs) - as a RAW copy in the caller's order.  The three agree: planes =
    own planes; raw = the caller's array, and permuting it with ob_orig (what the device kernel does) gives the planes; ob_orig, ob_cam and the
    count of kept residual blocks do not depend on the mode; NaN / Inf in the observations is refused in every mode."""
    rng = np.random.default_rng(2)
    for seed, nl, grouping in ((3, 150, 0), (4, 400, 1), (5, 60, 1)):
        w = synth.make_window(seed, num_lines=nl)
        if seed == 5:
            perm = rng.permutation(len(w["camera_index"]))
            for k in ("camera_index", "line_index", "observations"):
                w[k] = np.asarray(w[k])[perm]
            w["fixed_index"] = np.asarray(w["fixed_index"]).reshape(-1, 2)[perm].reshape(-1)
        Cn, L, M = int(w["num_cameras"]), int(w["num_lines"]), len(w["camera_index"])
        cam = np.ascontiguousarray(w["camera_index"], dtype=np.int32); line = np.ascontiguousarray(w["line_index"], dtype=np.int32)
        fixed = np.ascontiguousarray(w["fixed_index"], dtype=np.int32); obs = np.ascontiguousarray(w["observations"], dtype=np.float64).reshape(-1)
        prm = np.ascontiguousarray(w["parameters"], dtype=np.float64).copy()
        res = {}
        for mode in (0, 1, 2):
            planes, raw, own = np.full(8 * M, -7.0), np.full(8 * M, -7.0), np.zeros(8 * M)
            oo, oc, nk = np.zeros(M, np.int32), np.zeros(M, np.int32), C.c_int(0)
            rc = host_math.hm_pack_dest(Cn, L, M, _ip(cam), _ip(line), _ip(fixed), _dp(obs), _dp(prm), grouping, mode, _dp(planes), _dp(raw), _ip(oo), _ip(oc),
                                        C.byref(nk), _dp(own))
            assert rc == 0
            res[mode] = (planes, raw, own, oo, oc, nk.value)
        for mode in (1, 2):
            assert np.array_equal(res[mode][3], res[0][3]) and np.array_equal(res[mode][4], res[0][4]) and res[mode][5] == res[0][5]
        assert np.array_equal(res[1][0], res[0][2])                               # caller planes = the packed window's planes
        assert np.array_equal(res[2][1], obs) and np.all(res[2][0] == -7.0)        # raw mode: the caller's array, planes untouched
        gathered = obs.reshape(M, 4, 2)[res[2][3]]                                 # what k_permute_obs does: sorted position o <- observation ob_orig[o]
        assert np.array_equal(gathered.transpose(1, 0, 2).reshape(-1), res[0][2])
        bad = obs.copy(); bad[8 * (M // 2) + 3] = np.inf
        for mode in (0, 1, 2):
            planes, raw, own = np.zeros(8 * M), np.zeros(8 * M), np.zeros(8 * M)
            oo, oc, nk = np.zeros(M, np.int32), np.zeros(M, np.int32), C.c_int(0)
            assert host_math.hm_pack_dest(Cn, L, M, _ip(cam), _ip(line), _ip(fixed), _dp(bad), _dp(prm), grouping, mode, _dp(planes), _dp(raw), _ip(oo), _ip(oc),
                                          C.byref(nk), _dp(own)) == 1


def test_round6_entry_points_without_a_device():
    """The round-6 surface (include/slslam_hip.h): slslam_pack_indices is host arithmetic and works anywhere - the narrowed word is
    line | camera << 16 | camera constant << 24 | line constant << 25, indices beyond 255 cameras / 65534 lines are refused; page-locked
    memory, the device build's test hook and streams need a HIP device (SLSLAM_ERR_NO_DEVICE), never a CPU stand-in."""
    import ctypes as C
    from slslam_amd import capi
    L = capi.lib()
    w = synth.make_window(3, num_lines=40)
    cam = np.ascontiguousarray(w["camera_index"], dtype=np.int32); line = np.ascontiguousarray(w["line_index"], dtype=np.int32)
    fixed = np.ascontiguousarray(w["fixed_index"], dtype=np.int32)
    fixed[2 * 5 + 1] = 1
    pk = np.zeros(len(cam), dtype=np.uint32)
    assert L.slslam_pack_indices(len(cam), _ip(cam), _ip(line), _ip(fixed), pk.ctypes.data_as(C.POINTER(C.c_uint))) == 0
    assert np.array_equal(pk & 0xffff, line) and np.array_equal((pk >> 16) & 0xff, cam)
    assert np.array_equal((pk >> 24) & 1, fixed[0::2] != 0) and np.array_equal((pk >> 25) & 1, fixed[1::2] != 0) and not np.any(pk >> 26)
    bad = cam.copy(); bad[0] = 256
    assert L.slslam_pack_indices(len(cam), _ip(bad), _ip(line), _ip(fixed), pk.ctypes.data_as(C.POINTER(C.c_uint))) == 4
    bad = line.copy(); bad[0] = 65535
    assert L.slslam_pack_indices(len(cam), _ip(cam), _ip(bad), _ip(fixed), pk.ctypes.data_as(C.POINTER(C.c_uint))) == 4
    assert L.slslam_pack_indices(3, None, _ip(line), _ip(fixed), pk.ctypes.data_as(C.POINTER(C.c_uint))) == 1
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    p = C.c_void_p()
    assert L.slslam_pinned_alloc(4096, C.byref(p)) == 2 and not p.value
    buf = np.zeros(4096, dtype=np.uint8)
    assert L.slslam_pinned_register(buf.ctypes.data_as(C.c_void_p), 4096) == 2
    assert L.slslam_pinned_contains(buf.ctypes.data_as(C.c_void_p), 16) == 0
    with pytest.raises(capi.SlslamError) as e:
        capi.debug_device_pack(w)
    assert e.value.status == 2
    with pytest.raises(capi.SlslamError) as e:
        capi.LBAStream()
    assert e.value.status == 2
