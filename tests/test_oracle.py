"""CPU tests of the oracle (oracle/*.c): pinned against the committed golden fixtures
(tests/golden/, an independent numpy/scipy transcription) and the survey's known-answer vector.
The reference ships no tests or fixtures for this path."""
import os

import numpy as np
import pytest

from slslam_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: z[k] for k in z.files}


def test_survey_known_answer_vector(oracle):
    k = _load("residual_kat.npz")
    r = oracle.line_residual(k["survey_cam"], k["survey_line"], k["survey_obs"])
    assert np.abs(r - k["survey_residual"]).max() < 1e-15
    r0 = oracle.line_residual(np.zeros(6), k["survey_line"], k["survey_obs"])
    assert np.abs(r0 - k["survey_residual_cam0"]).max() < 1e-15
    s = float(k["survey_residual"] @ k["survey_residual"])
    rho = oracle.huber(s, 1.0 / 406.05)
    assert abs(s - 1.1061260053319433) < 1e-14
    assert abs(rho[0] - 5.17420946376693e-3) < 1e-15
    assert abs(rho[1] - 2.341629516328061e-3) < 1e-15


def test_line_residual_golden_vectors(oracle):
    k = _load("residual_kat.npz")
    for cam, line, obs, r_ref, jc_fd, jl_fd in zip(k["cam"], k["line"], k["obs"], k["residual"], k["j_cam_fd"], k["j_line_fd"]):
        r = oracle.line_residual(cam, line, obs)
        rj, jc, jl = oracle.line_residual_jet(cam, line, obs)
        assert np.abs(r - r_ref).max() < 5e-15
        assert np.abs(rj - r).max() < 1e-15
        # dual-number Jacobians against central differences of the independent transcription
        assert np.abs(jc - jc_fd).max() < 5e-8 * (1 + np.abs(jc).max())
        assert np.abs(jl - jl_fd).max() < 5e-8 * (1 + np.abs(jl).max())


def test_pose_residual_golden_vectors(oracle):
    k = _load("residual_kat.npz")
    for p1, p2, c, r_ref in zip(k["pose1"], k["pose2"], k["constraint"], k["pose_residual"]):
        r, j1, j2 = oracle.pose_residual_jet(p1, p2, c)
        assert np.abs(r - r_ref).max() < 1e-14
        h = 1e-6
        for k_ in range(6):
            dp = np.zeros(6); dp[k_] = h
            fd1 = (oracle.pose_residual_jet(p1 + dp, p2, c)[0] - oracle.pose_residual_jet(p1 - dp, p2, c)[0]) / (2 * h)
            fd2 = (oracle.pose_residual_jet(p1, p2 + dp, c)[0] - oracle.pose_residual_jet(p1, p2 - dp, c)[0]) / (2 * h)
            assert np.abs(j1[:, k_] - fd1).max() < 1e-7
            assert np.abs(j2[:, k_] - fd2).max() < 1e-7


def test_huber_loss(oracle):
    a = 1.0 / 406.05
    assert np.allclose(oracle.huber(0.5 * a * a, a), [0.5 * a * a, 1.0, 0.0])
    s = 9.0 * a * a
    rho = oracle.huber(s, a)
    assert abs(rho[0] - (2 * a * 3 * a - a * a)) < 1e-18
    assert abs(rho[1] - 1.0 / 3.0) < 1e-15
    assert rho[2] < 0


def test_orthonormal_round_trip(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        p = rng.normal(size=3) * 3
        cp = p - (p @ d) * d
        o = oracle.av_to_orth(np.concatenate([cp, d]))
        av = oracle.orth_to_av(o)
        assert np.abs(av[:3] - cp).max() < 1e-12 and np.abs(av[3:] - d).max() < 1e-12
        assert np.abs(o - synth.av_to_orth(np.concatenate([cp, d]))).max() < 1e-15
        assert np.abs(av - synth.orth_to_av(o)).max() < 1e-15


def test_consistent_segment_has_zero_residual(oracle):
    # a 3-D segment projected into both cameras and encoded with gc_av_to_orth gives |r| ~ 0
    w = synth.make_window(5, num_lines=40, noise_px=0.0)
    c = oracle.lba_cost(w, w["true_parameters"])
    assert c < 1e-25


def test_dense_and_schur_linear_solvers_agree(oracle):
    w = synth.make_window(7, num_lines=40, num_kf=8, num_free=4)
    xd, sd, td = oracle.lba_solve(w, linear_solver=0, max_num_iterations=3)
    xs, ss, ts = oracle.lba_solve(w, linear_solver=1, max_num_iterations=3)
    assert sd["num_free_parameters"] == 6 * 4 + 4 * 40
    for a, b in zip(td, ts):
        assert abs(a["cost"] - b["cost"]) <= 1e-10 * abs(a["cost"])
        assert a["step_is_successful"] == b["step_is_successful"]
    assert np.abs(xd - xs).max() < 1e-7


def test_lm_converges_to_scipy_optimum(oracle):
    g = _load("lba_optimum.npz")
    w = dict(g, num_cameras=int(g["num_cameras"]), num_lines=int(g["num_lines"]))
    assert abs(oracle.lba_cost(w, g["parameters"]) - float(g["initial_cost"])) < 1e-15
    assert abs(oracle.lba_cost(w, g["optimum"]) - float(g["optimum_cost"])) < 1e-16
    x, s, tr = oracle.lba_solve(w, max_num_iterations=200, function_tolerance=1e-16, parameter_tolerance=1e-14,
                                gradient_tolerance=1e-16)
    assert abs(s["final_cost"] - float(g["optimum_cost"])) < 1e-11 * float(g["optimum_cost"]) + 1e-16
    assert np.abs(x - g["optimum"]).max() < 1e-4
    # default policy: 10 iterations get within 1 % of the optimum cost
    x10, s10, _ = oracle.lba_solve(w)
    assert s10["final_cost"] < 1.01 * float(g["optimum_cost"])
    assert s10["num_successful_steps"] + s10["num_unsuccessful_steps"] <= 10


def test_constant_blocks_follow_ceres_semantics(oracle):
    w = synth.make_motion_only(2, num_lines=30)
    x, s, tr = oracle.lba_solve(w)
    C = w["num_cameras"]
    assert s["num_free_parameters"] == 6
    assert s["fixed_cost"] > 0                       # blocks with both parameter blocks constant
    assert np.array_equal(x[6:], w["parameters"][6:])  # camera 1 and every line untouched
    assert not np.array_equal(x[:6], w["parameters"][:6])
    assert s["final_cost"] <= s["initial_cost"]
    # everything constant -> nothing to do, costs are the fixed cost
    w2 = dict(w)
    w2["fixed_index"] = np.ones_like(w["fixed_index"])
    x2, s2, _ = oracle.lba_solve(w2)
    assert s2["num_free_parameters"] == 0 and s2["initial_cost"] == s2["final_cost"] == s2["fixed_cost"]
    assert np.array_equal(x2, w["parameters"])
    assert C == 2


def test_iteration_limit_counts_successful_and_unsuccessful(oracle):
    w = synth.make_window(9, num_lines=60)
    for k in (0, 1, 3):
        x, s, tr = oracle.lba_solve(w, linear_solver=1, max_num_iterations=k)
        assert s["num_successful_steps"] + s["num_unsuccessful_steps"] <= k
        assert len(tr) == 1 + s["num_successful_steps"] + s["num_unsuccessful_steps"]
    assert np.array_equal(oracle.lba_solve(w, linear_solver=1, max_num_iterations=0)[0], w["parameters"])


def test_pose_graph_consistent_is_zero_and_solves(oracle):
    g = synth.make_pose_graph(1, num_poses=30, num_loops=3)
    # a consistent graph: constraints from the true poses -> zero residual at the truth
    truth = g["true_parameters"].reshape(-1, 6)
    cons = []
    for a, b in zip(g["pose_index_1"], g["pose_index_2"]):
        Ra, ta = synth.wt_to_rt(truth[a]); Rb, tb = synth.wt_to_rt(truth[b])
        Rrel = Rb @ Ra.T
        cons.append(synth.rt_to_wt(Rrel, tb - Rrel @ ta))
    g0 = dict(g, constraints=np.array(cons))
    assert oracle.po_cost(g0, g["true_parameters"]) < 1e-25
    x, s, tr = oracle.po_solve(g)
    assert s["final_cost"] < s["initial_cost"]
    assert np.array_equal(x[:6], g["parameters"][:6])          # pose1 of edge 0 is the gauge
    p = _load("po_optimum.npz")
    gp = dict(p, num_poses=int(p["num_poses"]))
    xo, so, _ = oracle.po_solve(gp, max_num_iterations=100, function_tolerance=1e-16, parameter_tolerance=1e-14,
                                gradient_tolerance=1e-16)
    assert abs(so["final_cost"] - float(p["optimum_cost"])) < 1e-10 * float(p["optimum_cost"])
    assert np.abs(xo - p["optimum"]).max() < 1e-6


@pytest.mark.parametrize("seed", [0, 1])
def test_generator_is_deterministic_and_well_formed(seed):
    a = synth.make_window(seed, num_lines=100)
    b = synth.make_window(seed, num_lines=100)
    for k in ("camera_index", "line_index", "fixed_index", "observations", "parameters"):
        assert np.array_equal(a[k], b[k])
    M = len(a["camera_index"])
    assert a["observations"].shape == (M, 8) and len(a["fixed_index"]) == 2 * M
    assert len(a["parameters"]) == 6 * 20 + 4 * 100
    assert np.all(np.diff(a["line_index"]) >= 0)                       # grouped by line (slam.cpp:848-882)
    assert np.array_equal(a["parameters"][6 * 9:6 * 10], np.zeros(6))   # newest keyframe is identity
    free_obs = np.bincount(a["line_index"][a["camera_index"] < 10], minlength=100)
    assert free_obs.min() >= 2                                          # slam.cpp:839-840
    assert np.array_equal(a["fixed_index"][0::2], (a["camera_index"] >= 10).astype(np.int32))


def test_lm_trace_matches_independent_numpy_lm(oracle):
    """tests/golden/lba_lm_trace.npz: a Levenberg-Marquardt loop written in numpy from the Ceres 1.7.0 policy table on the
    numpy transcription of the residual (central-difference Jacobians, dense normal equations).  The oracle's
    trust-region bookkeeping (lm_core.c) must walk the same path: same accept / reject decisions, radii and costs."""
    z = np.load(os.path.join(GOLD, "lba_lm_trace.npz"))
    w = {k: z[k] for k in ("camera_index", "line_index", "fixed_index", "observations", "parameters")}
    w["num_cameras"], w["num_lines"] = int(z["num_cameras"]), int(z["num_lines"])
    x, s, tr = oracle.lba_solve(w, linear_solver=0)
    assert len(tr) == len(z["iteration"])
    assert int((z["successful"][1:] == 0).sum()) >= 1                   # the path has rejected steps to follow
    for k, rec in enumerate(tr):
        assert rec["iteration"] == int(z["iteration"][k]) and rec["step_is_successful"] == int(z["successful"][k])
        assert abs(rec["cost"] - z["cost"][k]) <= 2e-6 * z["cost"][k]     # finite-difference Jacobians on the numpy side
        assert abs(rec["trust_region_radius"] - z["radius"][k]) <= 1e-4 * z["radius"][k]
        if k:
            assert abs(rec["relative_decrease"] - z["relative_decrease"][k]) <= 5e-4 * max(1.0, abs(z["relative_decrease"][k]))
    assert np.abs(x - z["final_parameters"]).max() < 1e-5
    x1, s1, tr1 = oracle.lba_solve(w, linear_solver=1)                  # the Schur back-end walks it too
    assert [r["step_is_successful"] for r in tr1] == [int(v) for v in z["successful"]]


def test_po_envelope_cholesky_equals_dense(oracle):
    """linear_solver = 2 (envelope Cholesky in the natural pose order, the sparse stand-in for SPARSE_NORMAL_CHOLESKY at
    reference src/po_problem.cpp:68) solves the same normal equations as the dense factorisation: same steps, same poses."""
    for seed, n, loops in ((7, 120, 4), (3, 60, 0), (5, 90, 7)):
        g = synth.make_pose_graph(seed, num_poses=n, num_loops=loops)
        xd, sd, td = oracle.po_solve(g)
        xs, ss, ts = oracle.po_solve(g, linear_solver=2)
        assert (sd["num_successful_steps"], sd["num_unsuccessful_steps"]) == (ss["num_successful_steps"], ss["num_unsuccessful_steps"])
        assert abs(sd["final_cost"] - ss["final_cost"]) <= 1e-12 * max(sd["final_cost"], 1e-30) + 1e-20
        assert np.abs(xd - xs).max() < 1e-10
        assert len(td) == len(ts)
