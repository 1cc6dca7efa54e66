"""Plausibility against the only aggregates the reference publishes for the LBA path
(matlab_script/result_comp_ancdir_orthonorm/ba_result_orthonorm_*: BASELINE.md section 1) on a re-creation of its simulated run
(tools/house_study.py: 74-segment house model, circular "wave" trajectory, per-keyframe motion-only BA + sliding-window LBA).
The reference's simulator is not shipped, so the assertions are on the SCALING those files show, not on their digits; parity
of the GPU path on this pipeline is checked against the oracle run of the same seeds."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import house_study as hs  # noqa: E402


def test_house_model_restatement():
    segs = hs.house_segments()
    assert segs.shape == (74, 2, 3)
    assert segs[..., 0].min() == 0 and segs[..., 0].max() == 4.5 and segs[..., 2].max() == 3.5       # 4.5 x 4.5 x 3.5 m
    assert len({tuple(sorted([tuple(np.round(s[0], 9)), tuple(np.round(s[1], 9))])) for s in segs}) == 74   # no duplicate segment
    poses = hs.wave_trajectory(400)
    c = np.array([-(R.T @ t) for R, t in poses])
    r = np.hypot(c[:, 0] - 2.25, c[:, 1] - 2.25)
    assert np.allclose(r, 5.1) and abs(c[:, 2].max() - 2.0) < 1e-3 and abs(c[:, 2].min() - 1.0) < 1e-3   # radius, +-0.5 m wave
    step = np.linalg.norm(np.diff(c, axis=0), axis=1)
    assert 0.075 < step.min() and step.max() < 0.12              # the shipped trajectories: 0.079 .. 0.114 m per keyframe


def test_scaling_of_the_published_aggregates(oracle):
    solve = lambda w, it: oracle.lba_solve(w, linear_solver=1, max_num_iterations=it)[:2]
    runs = {(s, W): hs.run(s, W, solve, frames=170)["second_half"] for (s, W) in ((0.2, 5), (0.2, 10), (1.0, 5))}
    ref = hs.REFERENCE
    # cost proportional to the window size (reference: 1.0266e-3 / 4.978e-4 = 2.06)
    ratio_w = runs[(0.2, 10)]["avg_final_cost"] / runs[(0.2, 5)]["avg_final_cost"]
    assert abs(ratio_w / (ref[(0.2, 10)][2] / ref[(0.2, 5)][2]) - 1) < 0.2, ratio_w
    # noise 0.2 -> 1.0 px: x25 would be purely quadratic; the Huber knee at 1 px (1 / 406.05) bends it to x17.7 in the
    # reference's files - and here
    ratio_s = runs[(1.0, 5)]["avg_final_cost"] / runs[(0.2, 5)]["avg_final_cost"]
    assert abs(ratio_s / (ref[(1.0, 5)][2] / ref[(0.2, 5)][2]) - 1) < 0.15, ratio_s
    for key, r in runs.items():
        assert 0.4 < r["avg_final_cost"] / ref[key][2] < 1.2, (key, r)                 # same magnitude (reference's scene shows more)
        assert 1.0 <= r["avg_initial_cost"] / r["avg_final_cost"] < 1.12, (key, r)      # windows start within a few % of their optimum
        assert 1.0 < r["avg_iterations"] <= 10.0
    assert runs[(0.2, 10)]["avg_iterations"] < runs[(0.2, 5)]["avg_iterations"] < runs[(1.0, 5)]["avg_iterations"]   # falls with W, grows with noise
    assert runs[(0.2, 10)]["avg_initial_cost"] / runs[(0.2, 10)]["avg_final_cost"] < runs[(0.2, 5)]["avg_initial_cost"] / runs[(0.2, 5)]["avg_final_cost"]


@pytest.mark.gpu
def test_gpu_pipeline_matches_oracle_pipeline(hip, oracle):
    """The same simulated run - every window built from the results of the windows before it, motion-only BA and LBA through
    the C ABI - on the GPU and on the oracle: aggregates and trajectory agree (the windows are warm-started, so the solves
    converge and do not amplify round-off the way the far-from-optimum bench windows do)."""
    so = lambda w, it: oracle.lba_solve(w, linear_solver=1, max_num_iterations=it)[:2]
    sg = lambda w, it: hip.lba_solve(w, max_num_iterations=it)[:2]
    a, b = hs.run(0.4, 5, so, frames=60), hs.run(0.4, 5, sg, frames=60)
    assert abs(a["avg_iterations"] - b["avg_iterations"]) <= 0.05 * a["avg_iterations"]
    assert abs(a["avg_final_cost"] - b["avg_final_cost"]) <= 1e-5 * a["avg_final_cost"]
    assert np.abs(a["positions"] - b["positions"]).max() < 1e-4


@pytest.mark.gpu
def test_gpu_window_size_40(hip, oracle):
    """W = 40 of the reference's study (ba_window_size is a flag, src/main.cpp:22): windows of up to 80 keyframes, 40 of them
    free, house lines tracked through all of them (more than 64 observations per line) - the global-memory path of
    lba_big.h.  The oracle drives the simulated run; every window it solves is also solved on the GPU from the same input
    and compared (the first 40 windows have no fixed keyframe: their gauge is held by the LM damping alone, so only their
    cost is compared)."""
    seen = {"cams": 0, "obs": 0, "windows": 0, "gauged": 0}

    def solve(w, it):
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1, max_num_iterations=it)
        if w["num_cameras"] > 1:
            x1, s1, t1 = hip.lba_solve(w, max_num_iterations=it)
            seen["cams"] = max(seen["cams"], int(w["num_cameras"]))
            seen["obs"] = max(seen["obs"], int(np.bincount(w["line_index"]).max()))
            seen["windows"] += 1
            assert abs(s0["initial_cost"] - s1["initial_cost"]) <= 1e-11 * s0["initial_cost"]
            assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-5 * s0["final_cost"]
            if np.asarray(w["fixed_index"]).reshape(-1, 2)[:, 0].any():            # a fixed keyframe pins the gauge
                seen["gauged"] += 1
                assert s0["num_successful_steps"] == s1["num_successful_steps"] and s0["termination_type"] == s1["termination_type"]
                assert np.abs(x0 - x1).max() < 1e-5
        return x0, s0
    hs.run(0.4, 40, solve, frames=100)
    assert seen["cams"] == 80 and seen["obs"] > 64 and seen["gauged"] >= 55
