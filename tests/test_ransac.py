"""RANSAC hypothesis scoring: oracle checks on CPU, bit-exact GPU parity."""
import numpy as np
import pytest

from slslam_amd import synth


def test_oracle_scoring_semantics(oracle):
    poses, obs, lines, true_pose = synth.make_ransac_frame(3, num_lines=120, num_hypotheses=60, noise_px=0.3)
    sc, inl = oracle.ransac_score(poses, obs, lines)
    K = len(obs)
    assert sc[0] == K and inl[0].all()                          # the true motion: every line is an inlier at 0.3 px
    tn = np.linalg.norm(poses[:, 9:], axis=1)
    assert np.array_equal(sc == -1, tn > 1.0)                   # `if ( motion[j].t.norm() > 1 ) continue;`
    assert np.array_equal(sc[sc >= 0], inl[sc >= 0].sum(1))
    assert sc[sc >= 0].min() < K // 2                            # badly perturbed hypotheses lose their inliers
    # threshold is strict and in normalised units: nothing is an inlier at thr = 0
    sc0, _ = oracle.ransac_score(poses, obs, lines, error_thr=0.0)
    assert np.all(sc0[sc0 >= 0] == 0)
    # error of one pair against an independent numpy evaluation (double precision everywhere)
    R, t = true_pose[:9].reshape(3, 3), true_pose[9:]
    e = []
    for cam in range(2):
        tt = t - np.array([0.12 * cam, 0, 0])
        n = np.cross(R @ lines[0, :3] + tt, R @ lines[0, 3:])
        n = n / np.hypot(n[0], n[1])
        e += [abs(n @ np.r_[obs[0, 4 * cam:4 * cam + 2], 1]), abs(n @ np.r_[obs[0, 4 * cam + 2:4 * cam + 4], 1])]
    import ctypes as C
    lib = oracle.lib()
    lib.oracle_reprojection_error.restype = C.c_float
    dp = C.POINTER(C.c_double)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    got = lib.oracle_reprojection_error(f(obs[0]), f(R), f(t), f(lines[0]), C.c_double(0.12))
    assert abs(got - np.mean(e)) < 1e-6 * np.mean(e) + 1e-9      # float accumulation in the reference


@pytest.mark.gpu
@pytest.mark.parametrize("seed,K,H", [(1, 150, 256), (2, 37, 19), (3, 300, 1000), (4, 64, 64)])
def test_gpu_scores_are_bit_identical(hip, oracle, seed, K, H):
    poses, obs, lines, _ = synth.make_ransac_frame(seed, num_lines=K, num_hypotheses=H)
    s0, m0 = oracle.ransac_score(poses, obs, lines)
    s1, m1 = hip.ransac_score(poses, obs, lines)
    assert np.array_equal(s0, s1) and np.array_equal(m0, m1)     # integer outputs: exact
    # empty inputs
    s, m = hip.ransac_score(poses[:0], obs, lines)
    assert len(s) == 0
    s, m = hip.ransac_score(poses, obs[:0], lines[:0])
    assert np.array_equal(s, np.where(np.linalg.norm(poses[:, 9:], axis=1) > 1.0, -1, 0))


def test_oracle_motion_generator_recovers_small_motion(oracle):
    """SLAM::vo_angle_axis_approx (slam.cpp:433-574) restated: on noise-free correspondences of a small
    frame-to-frame motion the linearised solver returns that motion (to the linearisation error), which
    pins the sign conventions (-baseline at the call site, T = previous -> current)."""
    fr = synth.make_ransac_pair(1, num_lines=80, noise_px=0.0, outlier_frac=0.0, rot_deg=0.3)
    for tr in fr["samples"][:8]:
        n, pose = oracle.vo_angle_axis_approx(fr["obs0"][tr], fr["obs1"][tr])
        assert n == 1
        assert np.abs(pose[:9] - fr["true_pose"][:9]).max() < 2e-4
        assert np.abs(pose[9:] - fr["true_pose"][9:]).max() < 5e-3
    # degenerate sample (a zero-length observed segment -> zero image line): no solution
    o0 = fr["obs0"][fr["samples"][0]].copy()
    o0[0, 2:4] = o0[0, 0:2]
    n, _ = oracle.vo_angle_axis_approx(o0, fr["obs1"][fr["samples"][0]])
    assert n == 0


def test_oracle_ransac_motion_finds_the_inlier_set(oracle):
    fr = synth.make_ransac_pair(2, num_lines=150, noise_px=0.3, outlier_frac=0.25, num_trials=200)
    tc, best, pose, inl = oracle.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"])
    good = ~fr["outliers"]
    assert 0 < tc < 200                                          # the adaptive bound stopped the loop early
    assert best >= 0.9 * good.sum() and (inl & fr["outliers"]).sum() <= 0.1 * fr["outliers"].sum() + 2
    assert np.abs(pose[9:] - fr["true_pose"][9:]).max() < 0.05
    # the loop honours an incoming best score that nothing beats
    tc2, best2, _, _ = oracle.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"], best_score=10 ** 6)
    assert best2 == 10 ** 6 and tc2 == min(len(fr["obs0"]), 200)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,K,T", [(1, 150, 256), (2, 40, 64), (3, 300, 1001)])
def test_gpu_motion_generator_and_trial_loop(hip, oracle, seed, K, T):
    fr = synth.make_ransac_pair(seed, num_lines=K, noise_px=0.4, outlier_frac=0.3, num_trials=T)
    poses, valid = hip.ransac_generate(fr["obs0"], fr["obs1"], fr["samples"])
    assert valid.all()
    for h in range(0, T, max(1, T // 40)):
        n, ref = oracle.vo_angle_axis_approx(fr["obs0"][fr["samples"][h]], fr["obs1"][fr["samples"][h]])
        assert n == 1
        # fp64, same operation order; sin/cos of the device library differ from glibc by an ulp
        assert np.abs(poses[h] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    tc0, b0, p0, i0 = oracle.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"])
    tc1, b1, p1, i1 = hip.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"])
    assert (tc0, b0) == (tc1, b1)
    assert np.array_equal(i0, i1)
    assert np.abs(p0 - p1).max() < 1e-9
    # degenerate sample in trial 0 -> that trial is skipped (num_sol == 0), the loop goes on
    o0 = fr["obs0"].copy()
    k = fr["samples"][0][0]
    o0[k, 2:4] = o0[k, 0:2]
    _, v = hip.ransac_generate(o0, fr["obs1"], fr["samples"])
    uses = (fr["samples"] == k).any(axis=1)
    assert np.array_equal(v == 0, uses)
    t0 = oracle.ransac_motion(o0, fr["obs1"], fr["lines"], fr["samples"])
    t1 = hip.ransac_motion(o0, fr["obs1"], fr["lines"], fr["samples"])
    assert t0[:2] == t1[:2] and np.array_equal(t0[3], t1[3])


@pytest.mark.gpu
def test_gpu_ransac_motion_batch_equals_single_calls(hip, oracle):
    """slslam_ransac_motion_batch: many frames in one call give, per frame, what slslam_ransac_motion and the
    sequential oracle loop give (ragged frames, an empty one included)."""
    frames = [synth.make_ransac_pair(40 + i, num_lines=30 + 17 * i, noise_px=0.4, outlier_frac=0.3, num_trials=50 + 31 * i) for i in range(6)]
    empty = dict(frames[0], samples=frames[0]["samples"][:0])
    batch = hip.ransac_motion_batch(frames + [empty])
    assert len(batch) == 7 and batch[6][0] == 0 and batch[6][1] == 0
    for fr, (tc, best, pose, mask) in zip(frames, batch[:6]):
        t1 = hip.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"])
        t0 = oracle.ransac_motion(fr["obs0"], fr["obs1"], fr["lines"], fr["samples"])
        assert (tc, best) == t1[:2] == t0[:2]
        assert np.array_equal(mask, t1[3]) and np.array_equal(mask, t0[3])
        assert np.array_equal(pose, t1[2])
