"""Plausibility study against the only numbers the reference publishes for this path:
matlab_script/result_comp_ancdir_orthonorm/ba_result_orthonorm_err{E}_basize{B}_maxnumiter10.txt - average LM iterations per
frame and average initial / final LBA cost of a simulated ~400-keyframe run around the 74-segment "house" model
(matlab_script/house.m:33-133), for noise E in {0.2 .. 1.0} px and window sizes B in {5, 10, 20, 40}.

The reference's simulator (observation generator, ground-truth file) is NOT shipped, so this is a re-creation, not a replay:
  * scene: the 74 segments of the house model, restated below from its dimensions (4.5 x 4.5 x 3.5 m, roof from 65 % of
    the height, door / window rectangles, wall and roof subdivisions, wall diagonals);
  * trajectory: 400 keyframes on the circle of radius 5.1 m around the house with the +-0.5 m, two-period height wave of the
    shipped estimated trajectories (trajectory_orthonorm_err0.2_basize40_maxnumiter10.txt: x in [0, 10.2], z in [-5.05, 5.13],
    y in [-0.5, 0.5], yaw 0 .. 2 pi), the stereo rig (f = 406.05, 640 x 480, baseline 0.12: src/parameter.h:43-52) looking at
    the house;
  * pipeline per keyframe, mirroring SLAM's data flow (src/slam.cpp:223-319, 730-761, 1370-1427): pose prediction, motion-only
    BA against the mapped lines (LBAProblem, one free camera, constant lines), new landmarks by stereo triangulation
    (initialize_lm), sliding-window LBA over the 2 W newest keyframes (W free + W fixed) with max 10 iterations, write-back.
What can be asserted is the SCALING the 40 files show, not their digits: cost proportional to W, to sigma^2 below the Huber
knee, initial cost within a few % of the final one, 2-6 LM iterations per frame falling with W (tests/test_house_study.py).

    python tools/house_study.py [--backend oracle|hip] [--frames 400] [--sigmas 0.2,0.6,1.0] [--windows 5,10,20,40]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from slslam_amd import synth  # noqa: E402

# the reference's published aggregates, orthonormal parametrisation, max_num_iter = 10:
# (sigma px, W) -> (avg LM iterations per frame, avg initial cost, avg final cost)
REFERENCE = {(0.2, 5): (2.33582, 5.33873e-4, 4.97805e-4), (0.2, 10): (2.21393, 1.04857e-3, 1.02656e-3),
             (0.2, 20): (1.4602, 2.06725e-3, 2.05101e-3), (0.2, 40): (1.13682, 3.9544e-3, 3.94087e-3),
             (0.6, 10): (4.27114, 8.47761e-3, 8.38284e-3),
             (1.0, 5): (7.49254, 9.09799e-3, 8.79131e-3), (1.0, 10): (5.34826, 1.8233e-2, 1.80694e-2),
             (1.0, 20): (4.05224, 3.59655e-2, 3.5877e-2), (1.0, 40): (3.24129, 6.86576e-2, 6.86017e-2)}


def house_segments(l=4.5, w=4.5, h=3.5):
    """The 74 segments of the house model as (74, 2, 3) endpoints, house frame: x along the length, y along the width, z up."""
    r, q, p = 0.65, 0.5, 0.25                    # roof starts at r h; door / window top at q h; window sill at p h
    a, b, c, d = 0.2, 0.4, 0.6, 0.8             # window spans a w .. b w, door c w .. d w (on the wall x = 0)
    S = []
    seg = lambda A, B: S.append((np.array(A, float), np.array(B, float)))
    rect_yz = lambda x, y0, y1, z0, z1: [seg((x, y0, z0), (x, y1, z0)), seg((x, y1, z0), (x, y1, z1)),
                                         seg((x, y1, z1), (x, y0, z1)), seg((x, y0, z1), (x, y0, z0))]
    for (x, y) in ((0, 0), (l, 0), (l, w), (0, w)):                                   # 1-4 wall corners
        seg((x, y, 0), (x, y, r * h))
    seg((0, 0, 0), (l, 0, 0)); seg((l, 0, 0), (l, w, 0)); seg((l, w, 0), (0, w, 0)); seg((0, w, 0), (0, 0, 0))   # 5-8 floor
    for x in (0, l):                                                                   # 9-12 gable slopes
        seg((x, 0, r * h), (x, w / 2, h)); seg((x, w / 2, h), (x, w, r * h))
    seg((0, w / 2, h), (l, w / 2, h)); seg((0, 0, r * h), (l, 0, r * h)); seg((0, w, r * h), (l, w, r * h))      # 13-15 ridge, eaves
    rect_yz(0, c * w, d * w, 0, q * h)                                                 # 16-19 door
    rect_yz(0, a * w, b * w, p * h, q * h)                                             # 20-23 window
    seg((0, 0, r * h), (0, w, r * h)); seg((l, 0, r * h), (l, w, r * h))               # 24-25 gable bases
    seg((0, a * w, (p + q) * h / 2), (0, b * w, (p + q) * h / 2)); seg((0, (a + b) * w / 2, p * h), (0, (a + b) * w / 2, q * h))   # 26-27 window cross
    for x in (l / 2, l / 4, 3 * l / 4):                                                # 28-30 rafters, front slope
        seg((x, 0, r * h), (x, w / 2, h))
    for x in (l / 2, l / 4, 3 * l / 4):                                                # 31-33 rafters, back slope
        seg((x, w / 2, h), (x, w, r * h))
    for k in (1, 2, 3):                                                                # 34-36 purlins, front slope
        seg((0, w * k / 8, r * h + (h - r * h) * k / 4), (l, w * k / 8, r * h + (h - r * h) * k / 4))
    for k, m in ((5, 3), (6, 2), (7, 1)):                                              # 37-39 purlins, back slope
        seg((0, w * k / 8, r * h + (h - r * h) * m / 4), (l, w * k / 8, r * h + (h - r * h) * m / 4))
    for y in (0, w):                                                                   # 40-45 wall posts, long walls
        for k in (1, 2, 3):
            seg((l * k / 4, y, 0), (l * k / 4, y, r * h))
    for k in (1, 2, 3):                                                                # 46-48 wall posts, far wall
        seg((l, w * k / 4, 0), (l, w * k / 4, r * h))
    seg((0, c * w, 0), (0, d * w, q * h)); seg((0, d * w, 0), (0, c * w, q * h))       # 49-50 door diagonals
    for k in range(4):                                                                 # 51-58 diagonals, wall y = 0
        seg((l * k / 4, 0, 0), (l * (k + 1) / 4, 0, r * h)); seg((l * (k + 1) / 4, 0, 0), (l * k / 4, 0, r * h))
    for k in range(4):                                                                 # 59-66 diagonals, wall x = l
        seg((l, w * k / 4, 0), (l, w * (k + 1) / 4, r * h)); seg((l, w * (k + 1) / 4, 0), (l, w * k / 4, r * h))
    for k in range(4):                                                                 # 67-74 diagonals, wall y = w
        seg((l * k / 4, w, 0), (l * (k + 1) / 4, w, r * h)); seg((l * (k + 1) / 4, w, 0), (l * k / 4, w, r * h))
    out = np.array([[A, B] for A, B in S])
    assert out.shape == (74, 2, 3)
    return out


def wave_trajectory(n=400, radius=5.1, amp=0.5, house_centre=(2.25, 2.25, 1.75), cam_height=1.5):
    """World -> camera poses (R, t) of n keyframes circling the house, camera x right / y down / z forward, looking at the
    house; world frame = house frame (z up)."""
    poses = []
    for k in range(n):
        th = 2 * np.pi * k / n
        c = np.array([house_centre[0] - radius * np.cos(th), house_centre[1] - radius * np.sin(th), cam_height + amp * np.sin(2 * th)])
        fwd = np.array(house_centre) - c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])                 # rows: camera axes in the world
        poses.append((R, -R @ c))
    return poses


def observe(R, t, segs, sigma_px, rng, visibility="both"):
    """Stereo observation [8] (normalised, as SLAM::insert_curr_obs) of every segment fully inside both images
    (visibility "both"), inside the left image only ("left") or merely in front of the camera ("front")."""
    ids, obs = [], []
    for i, (A, B) in enumerate(segs):
        o, ok = np.empty(8), True
        for e, P in enumerate((A, B)):
            ul, vl, ur, z = synth._project(R, t, P)
            if visibility == "both":
                ok &= bool(z > 0.3 and 0 <= ul < synth.WIDTH and 0 <= ur < synth.WIDTH and 0 <= vl < synth.HEIGHT)
            elif visibility == "left":
                ok &= bool(z > 0.3 and 0 <= ul < synth.WIDTH and 0 <= vl < synth.HEIGHT)
            else:
                ok &= bool(z > 0.3)
            o[2 * e], o[2 * e + 1], o[4 + 2 * e], o[4 + 2 * e + 1] = ul, vl, ur, vl
        if ok:
            o = o + rng.normal(0, sigma_px, 8)
            o[0::2] = (o[0::2] - synth.CX) / synth.FOCAL
            o[1::2] = (o[1::2] - synth.CY) / synth.FOCAL
            ids.append(i); obs.append(o)
    return ids, np.array(obs).reshape(-1, 8)


def run(sigma_px, W, solve, frames=400, seed=4, max_iter=10, lm_init="stereo", frames_per_keyframe=1, visibility="both",
        noise_scale=1.0, pose_noise=(0.2, 0.005), motion_only=True):
    """One simulated run; `solve(window_dict, max_num_iterations)` -> (parameters, summary).  Returns the averages the
    reference prints (src/main.cpp:84-89: sums over LBA calls divided by the frame count) and the estimated positions.

    The simulator unknowns (the reference's observation generator is not shipped), swept by --sweep:
      lm_init              "stereo": a new landmark is triangulated from its first noisy stereo observation (initialize_lm);
                           "truth": the true line with a small perturbation (1 mm / 0.05 deg on its end points)
      frames_per_keyframe  the reference divides its sums by frame_id (src/main.cpp:84-89), the number of FRAMES read; only
                           keyframes run the LBA (main.cpp:61-69).  k > 1 models a sequence with k frames per keyframe:
                           the same keyframes, denominators k times larger
      visibility           which segments a keyframe observes (see observe)
      noise_scale          the file name's noise level times this (a generator that perturbs, e.g., both the end points and the
                           stereo match adds more than sigma per coordinate)
      pose_noise           (deg, m) perturbation of the predicted pose that stands in for the visual odometry
      motion_only          run motion-only BA on the predicted pose before the window LBA (pose_estimation, slam.cpp:303)"""
    rng = np.random.default_rng([seed, int(round(100 * sigma_px)), W])
    sigma_nominal = sigma_px
    sigma_px = sigma_px * noise_scale
    segs = house_segments()
    truth = wave_trajectory(frames)
    R0, t0 = truth[0]
    to_first = lambda R, t: (R @ R0.T, t - R @ R0.T @ t0)          # express poses relative to keyframe 0 (the map's origin)
    truth = [to_first(R, t) for R, t in truth]
    segs0 = np.einsum("ij,nej->nei", R0, segs) + t0                # segments in the frame of keyframe 0
    kf_pose, kf_obs, lines = [], [], {}                             # estimated (w,t)[6]; {line id: obs[8]}; world orth[4]
    s_it = s_c0 = s_c1 = 0.0
    per_frame = []
    term_hist, iter_hist = {}, {}
    t_mo = t_lba = 0.0                                              # seconds inside the solver calls (motion-only / window LBA)
    n_mo = n_lba = obs_lba = 0
    for k in range(frames):
        ids, ob = observe(truth[k][0], truth[k][1], segs0, sigma_px, rng, visibility)
        obs_k = dict(zip(ids, ob))
        # ---- pose prediction: last estimate composed with the true relative motion, perturbed (stands in for RANSAC VO)
        if k == 0:
            pose = np.zeros(6)
        else:
            Rp, tp = synth.wt_to_rt(kf_pose[-1])
            Rrel = truth[k][0] @ truth[k - 1][0].T
            trel = truth[k][1] - Rrel @ truth[k - 1][1]
            Rn = synth.rodrigues(rng.normal(0, np.deg2rad(pose_noise[0]), 3)) @ Rrel @ Rp
            pose = synth.rt_to_wt(Rn, Rrel @ tp + trel + rng.normal(0, pose_noise[1], 3))
            # ---- motion-only BA against the mapped lines (SLAM::motion_only_ba: camera 0 free, lines constant)
            common = [i for i in ids if i in lines]
            if len(common) >= 5 and motion_only:
                m = len(common)
                w = {"num_cameras": 1, "num_lines": m, "camera_index": np.zeros(m, np.int32), "line_index": np.arange(m, dtype=np.int32),
                     "fixed_index": np.tile([0, 1], m).astype(np.int32), "observations": np.array([obs_k[i] for i in common]),
                     "parameters": np.concatenate([pose, np.concatenate([lines[i] for i in common])])}
                tq = time.perf_counter()
                x, _ = solve(w, max_iter)
                t_mo += time.perf_counter() - tq; n_mo += 1
                pose = x[:6]
        kf_pose.append(pose); kf_obs.append(obs_k)
        # ---- new landmarks: stereo triangulation in the keyframe, moved to the map frame (initialize_lm, gc_line_from_pose)
        R, t = synth.wt_to_rt(pose)
        for i in ids:
            if i not in lines:
                if lm_init == "stereo":
                    lm = synth._initialize_lm(obs_k[i][None])[0]
                    lines[i] = synth.av_to_orth(np.concatenate([R.T @ (lm[:3] - t), R.T @ lm[3:]])[None])[0]
                else:
                    A, B = segs0[i][0] + rng.normal(0, 1e-3, 3), segs0[i][1] + rng.normal(0, 1e-3, 3)
                    dv = (B - A) / np.linalg.norm(B - A)
                    cp = A - dv * (A @ dv)                                   # closest point of the line to the map origin
                    lines[i] = synth.av_to_orth(np.concatenate([cp, dv])[None])[0]
        # ---- sliding-window LBA: the 2 W newest keyframes, the W newest free (src/slam.cpp:1376-1382, 811-832)
        if k == 0:
            continue
        win = list(range(max(0, k - 2 * W + 1), k + 1))
        free = win[-W:]
        fixed = win[:-W]
        seen_free = {}
        for f in free:
            for i in kf_obs[f]:
                seen_free[i] = seen_free.get(i, 0) + 1
        lm_ids = sorted(i for i, c in seen_free.items() if c >= 2)                # slam.cpp:839-840
        if not lm_ids:
            continue
        lpos = {i: n for n, i in enumerate(lm_ids)}
        cams = free + fixed
        ci, li, fi, oo = [], [], [], []
        for i in lm_ids:
            for cpos, f in enumerate(cams):
                if i in kf_obs[f]:
                    ci.append(cpos); li.append(lpos[i]); fi += [1 if cpos >= len(free) else 0, 0]; oo.append(kf_obs[f][i])
        w = {"num_cameras": len(cams), "num_lines": len(lm_ids), "camera_index": np.array(ci, np.int32), "line_index": np.array(li, np.int32),
             "fixed_index": np.array(fi, np.int32), "observations": np.array(oo),
             "parameters": np.concatenate([np.concatenate([kf_pose[f] for f in cams]), np.concatenate([lines[i] for i in lm_ids])])}
        tq = time.perf_counter()
        x, s = solve(w, max_iter)
        t_lba += time.perf_counter() - tq; n_lba += 1; obs_lba += len(ci)
        for cpos, f in enumerate(free):
            kf_pose[f] = x[6 * cpos:6 * cpos + 6]
        for i in lm_ids:
            lines[i] = x[6 * len(cams) + 4 * lpos[i]:][:4]
        s_it += s["num_successful_steps"] + s["num_unsuccessful_steps"]
        s_c0 += s["initial_cost"]; s_c1 += s["final_cost"]
        tt = int(s.get("termination_type", -1))
        term_hist[tt] = term_hist.get(tt, 0) + 1
        ni = s["num_successful_steps"] + s["num_unsuccessful_steps"]
        iter_hist[ni] = iter_hist.get(ni, 0) + 1
        per_frame.append((s["num_successful_steps"] + s["num_unsuccessful_steps"], s["initial_cost"], s["final_cost"], s["num_unsuccessful_steps"]))
    # positions relative to the (estimated) first keyframe, as SLAM::save_trajectory re-roots them (metric_embedding(0))
    Re0, te0 = synth.wt_to_rt(kf_pose[0])
    est = []
    for pz in kf_pose:
        Rk, tk = synth.wt_to_rt(pz)
        c_map = -(Rk.T @ tk)                       # camera centre in map coordinates
        est.append(Re0 @ c_map + te0)              # ... in the frame of keyframe 0
    est = np.array(est)
    gt = np.array([-(R.T @ t) for R, t in truth])  # truth is already relative to keyframe 0
    err = np.linalg.norm(est - gt, axis=1)
    half = len(per_frame) // 2
    tail = per_frame[half:]
    denom = frames * frames_per_keyframe                  # frame_id of src/main.cpp:84-89
    return {"sigma_px": sigma_nominal, "W": W, "frames": frames, "avg_iterations": s_it / denom, "avg_initial_cost": s_c0 / denom,
            "avg_final_cost": s_c1 / denom, "mean_position_error_m": float(err.mean()),
            "per_call": {"iterations": s_it / max(1, n_lba), "initial_cost": s_c0 / max(1, n_lba), "final_cost": s_c1 / max(1, n_lba)},
            # 0 NO_CONVERGENCE (iteration cap), 1 GRADIENT_TOLERANCE, 2 FUNCTION_TOLERANCE, 3 PARAMETER_TOLERANCE, 4 NUMERICAL_FAILURE
            "termination_histogram": {str(k): v for k, v in sorted(term_hist.items())},
            "iterations_histogram": {str(k): v for k, v in sorted(iter_hist.items())},
            "settings": {"lm_init": lm_init, "frames_per_keyframe": frames_per_keyframe, "visibility": visibility,
                         "noise_scale": noise_scale, "pose_noise": list(pose_noise), "motion_only": motion_only},
            "solver_time": {"lba_ms_per_call": 1e3 * t_lba / max(1, n_lba), "lba_calls": n_lba, "avg_observations_per_window": obs_lba / max(1, n_lba),
                            "motion_only_ms_per_call": 1e3 * t_mo / max(1, n_mo), "motion_only_calls": n_mo,
                            "optimisation_ms_per_keyframe": 1e3 * (t_lba + t_mo) / frames},
            "second_half": {"avg_iterations": float(np.mean([q[0] for q in tail])), "avg_initial_cost": float(np.mean([q[1] for q in tail])),
                            "avg_final_cost": float(np.mean([q[2] for q in tail])), "rejected_step_fraction": float(np.sum([q[3] for q in tail]) / max(1, np.sum([q[0] for q in tail])))},
            "positions": est}


# ---------------------------------------------------------------------------------------------------------------------------
# The same simulated run driven through the REFERENCE's data flow around the two solves (numpy restatement of the glue the host
# library implements in C++: slslam_amd/host/window_packer.cpp, gc_lite.cpp):
#   * the visual odometry hands over the RELATIVE motion previous keyframe -> current frame; motion-only BA refines it with the
#     two-camera problem of SLAM::motion_only_ba (src/slam.cpp:578-675: camera 0 = the motion, camera 1 = identity and constant,
#     lines constant, given in the previous keyframe's frame); the new keyframe's pose is gc_T_20(motion, previous pose);
#   * a landmark keeps its line as (point, direction) in the frame of the keyframe that initialised it (lm->line, init_kf);
#   * the window arrays follow SLAM::bundle_adjustment (src/slam.cpp:811-921): free keyframes in ascending id, landmarks that are
#     members of >= 2 free keyframes in ascending id, each with all its observations in ba_kfs in chronological order, keyframes
#     of rank >= W appended as constant cameras when first met; write-back as src/slam.cpp:957-972.
# tests/host_cxx/house_replay.cpp does the same with the C++ host library from the scene file `dump` writes;
# tests/test_host_cxx.py::test_house_replay_cxx_equals_python compares the two trajectories.
def _line_to_pose(line, R, t):
    return np.concatenate([R @ line[:3] + t, R @ line[3:]])


def _line_from_pose(line, R, t):
    return _line_to_pose(line, R.T, -(R.T @ t))


def run_reference_protocol(sigma_px, W, solve, frames=400, seed=4, max_iter=10, dump=None):
    rng = np.random.default_rng([seed, int(round(100 * sigma_px)), W])
    segs = house_segments()
    truth = wave_trajectory(frames)
    R0, t0 = truth[0]
    truth = [(R @ R0.T, t - R @ R0.T @ t0) for R, t in truth]
    segs0 = np.einsum("ij,nej->nei", R0, segs) + t0
    kfT, members, lms, prev_obs = [], [], {}, None
    s_it = s_c0 = s_c1 = 0.0
    n_lba = n_mo = 0
    digests = []

    def fnv1a(data, h=1469598103934665603):
        for b in data:
            h = ((h ^ b) * 1099511628211) & 0xffffffffffffffff
        return h
    f = open(dump, "wb") if dump else None
    if f:
        np.array([frames, W, max_iter], dtype=np.int32).tofile(f)
    for k in range(frames):
        ids, ob = observe(truth[k][0], truth[k][1], segs0, sigma_px, rng)
        obs_k = dict(zip(ids, ob))
        tri = synth._initialize_lm(ob) if len(ids) else np.zeros((0, 6))          # stereo triangulation in the frame (initialize_lm)
        motion_wt = np.zeros(6)
        if k > 0:
            Rrel = truth[k][0] @ truth[k - 1][0].T
            trel = truth[k][1] - Rrel @ truth[k - 1][1]
            motion_wt = synth.rt_to_wt(synth.rodrigues(rng.normal(0, np.deg2rad(0.2), 3)) @ Rrel, trel + rng.normal(0, 0.005, 3))
        if f:
            np.array([len(ids)], dtype=np.int32).tofile(f)
            np.asarray(ids, dtype=np.int32).tofile(f)
            np.asarray(ob, dtype=np.float64).tofile(f)
            np.asarray(tri, dtype=np.float64).tofile(f)
            motion_wt.tofile(f)
        if k == 0:
            R, t = np.eye(3), np.zeros(3)
        else:
            Rm, tm = synth.wt_to_rt(motion_wt)
            Rp, tp = kfT[-1]
            common = [i for i in ids if i in lms and i in prev_obs]
            if len(common) >= 5:                                                    # SLAM::motion_only_ba
                m = len(common)
                lines_prev = [_line_to_pose(_line_from_pose(lms[i]["line"], *kfT[lms[i]["init"]]), Rp, tp) for i in common]
                w = {"num_cameras": 2, "num_lines": m, "camera_index": np.tile([0, 1], m).astype(np.int32),
                     "line_index": np.repeat(np.arange(m, dtype=np.int32), 2), "fixed_index": np.tile([0, 1, 1, 1], m).astype(np.int32),
                     "observations": np.array([[obs_k[i], prev_obs[i]] for i in common]).reshape(2 * m, 8),
                     "parameters": np.concatenate([synth.rt_to_wt(Rm, tm), synth.rt_to_wt(np.eye(3), np.zeros(3)),
                                                   synth.av_to_orth(np.array(lines_prev)).reshape(-1)])}
                x, _ = solve(w, max_iter)
                Rm, tm = synth.wt_to_rt(x[:6])
                n_mo += 1
            R, t = Rm @ Rp, Rm @ tp + tm                                            # gc_T_20(motion, previous pose)
        kfT.append((R, t)); members.append(list(ids))
        for n, i in enumerate(ids):
            if i not in lms:
                lms[i] = {"line": tri[n].copy(), "init": k, "obs": []}
            lms[i]["obs"].append((k, obs_k[i]))
        prev_obs = obs_k
        if k == 0:
            continue
        # ---- SLAM::bundle_adjustment on ba_kfs = the 2 W newest keyframes, rank = distance from the newest
        rank = lambda j: (k - j) if k - j < 2 * W else -1
        free = [j for j in range(k + 1) if 0 <= rank(j) < W]
        count = {}
        for j in free:
            for i in members[j]:
                count[i] = count.get(i, 0) + 1
        slot = {j: n for n, j in enumerate(free)}
        cam_kf = list(free)
        ci, li, fi, oo, lm_ids = [], [], [], [], []
        for i in sorted(count):
            if count[i] < 2:
                continue
            for (j, o8) in lms[i]["obs"]:
                if rank(j) < 0:
                    continue
                if j not in slot:
                    slot[j] = len(cam_kf); cam_kf.append(j)
                ci.append(slot[j]); li.append(len(lm_ids)); fi += [0 if slot[j] < W else 1, 0]; oo.append(o8)
            lm_ids.append(i)
        if not lm_ids:
            continue
        cams = np.concatenate([synth.rt_to_wt(*kfT[j]) for j in cam_kf])
        lines_w = np.array([_line_from_pose(lms[i]["line"], *kfT[lms[i]["init"]]) for i in lm_ids])
        w = {"num_cameras": len(cam_kf), "num_lines": len(lm_ids), "camera_index": np.array(ci, np.int32), "line_index": np.array(li, np.int32),
             "fixed_index": np.array(fi, np.int32), "observations": np.array(oo), "parameters": np.concatenate([cams, synth.av_to_orth(lines_w).reshape(-1)])}
        h = fnv1a(w["fixed_index"].tobytes(), fnv1a(w["line_index"].tobytes(), fnv1a(w["camera_index"].tobytes())))
        digests.append((k, len(cam_kf) | (len(lm_ids) << 16) | (len(ci) << 32), h, fnv1a(np.ascontiguousarray(w["observations"]).tobytes())))
        x, s = solve(w, max_iter)
        n_lba += 1
        for c, j in enumerate(cam_kf):                                              # slam.cpp:957-962: every camera, the constant ones too
            kfT[j] = synth.wt_to_rt(x[6 * c:6 * c + 6])
        for n, i in enumerate(lm_ids):                                              # :964-972: back into the (updated) initial keyframe
            lms[i]["line"] = _line_to_pose(synth.orth_to_av(x[6 * len(cam_kf) + 4 * n:][:4]), *kfT[lms[i]["init"]])
        s_it += s["num_successful_steps"] + s["num_unsuccessful_steps"]
        s_c0 += s["initial_cost"]; s_c1 += s["final_cost"]
    if f:
        f.close()
    poses = np.array([np.concatenate([R.reshape(-1), t]) for R, t in kfT])
    return {"poses": poses, "window_digests": digests, "lm_iterations": int(s_it), "sum_initial_cost": s_c0, "sum_final_cost": s_c1, "lba_calls": n_lba, "motion_only_calls": n_mo}



def sweep(solve, frames):
    """Which simulator setting, if any, reproduces BOTH published columns (LM iterations per frame and cost per frame) of
    the sigma = 0.2 and 1.0 px rows?  One factor at a time around the base setting, then the combinations the single factors
    point to.  Prints one line per (setting, sigma, W) with the ratios to the reference file."""
    base = dict(lm_init="stereo", frames_per_keyframe=1, visibility="both", noise_scale=1.0, pose_noise=(0.2, 0.005), motion_only=True)
    variants = [("base", {}), ("lm_init=truth", dict(lm_init="truth")), ("visibility=left", dict(visibility="left")),
                ("visibility=front", dict(visibility="front")), ("no motion-only BA", dict(motion_only=False)),
                ("pose prediction exact", dict(pose_noise=(0.0, 0.0))), ("pose prediction 5x worse", dict(pose_noise=(1.0, 0.025))),
                ("noise x1.36", dict(noise_scale=1.36)), ("noise x1.36, frames/kf=2", dict(noise_scale=1.36 * 2 ** 0.5, frames_per_keyframe=2)),
                ("noise x1.36, frames/kf=3", dict(noise_scale=1.36 * 3 ** 0.5, frames_per_keyframe=3)),
                ("frames/kf=3", dict(frames_per_keyframe=3)),
                ("truth init, noise x1.36", dict(lm_init="truth", noise_scale=1.36)),
                ("truth init, front, noise x1.25", dict(lm_init="truth", visibility="front", noise_scale=1.25))]
    print("%-34s %5s %3s | %8s %8s %6s | %9s %9s %6s | %5s | %s" % ("setting", "sigma", "W", "it/frame", "file", "ratio", "cost/frm", "file", "ratio", "it/call", "termination (0 cap, 2 function tol.) / iterations histogram"))
    for name, kw in variants:
        for sg in (0.2, 1.0):
            for W in (5, 10, 20):
                r = run(sg, W, solve, frames=frames, **{**base, **kw})
                ref = REFERENCE[(sg, W)]
                print("%-34s %5.1f %3d | %8.3f %8.3f %6.2f | %9.3e %9.3e %6.2f | %5.2f | %s %s" % (
                    name, sg, W, r["avg_iterations"], ref[0], r["avg_iterations"] / ref[0], r["avg_final_cost"], ref[2],
                    r["avg_final_cost"] / ref[2], r["per_call"]["iterations"], r["termination_histogram"], r["iterations_histogram"]), flush=True)


def function_tolerance_study(frames):
    """The one LM-policy number the iteration column is sensitive to: Ceres' function_tolerance (default 1e-6, which the reference
    does not override: src/lba_problem.cpp:95-132).  Oracle runs of the base setting with 1e-6 .. 1e-3."""
    from oracle import pyoracle
    for ftol in (1e-6, 1e-5, 1e-4, 1e-3):
        solve = lambda w, it: pyoracle.lba_solve(w, linear_solver=1, max_num_iterations=it, function_tolerance=ftol)[:2]
        out = []
        for sg, W in ((0.2, 5), (0.2, 10), (0.2, 20), (1.0, 5), (1.0, 10), (1.0, 20)):
            r = run(sg, W, solve, frames=frames)
            ref = REFERENCE[(sg, W)]
            out.append("%.1f/%d it %.2f (file %.2f) cost %.2f" % (sg, W, r["avg_iterations"], ref[0], r["avg_final_cost"] / ref[2]))
        print("function_tolerance %g: " % ftol + " | ".join(out), flush=True)


def make_solver(backend):
    if backend == "oracle":
        from oracle import pyoracle
        return lambda w, it: pyoracle.lba_solve(w, linear_solver=1, max_num_iterations=it)[:2]
    from slslam_amd import capi
    return lambda w, it: capi.lba_solve(w, max_num_iterations=it)[:2]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--sigmas", default="0.2,0.6,1.0")
    ap.add_argument("--windows", default="5,10,20,40")
    ap.add_argument("--dump-scene", default=None, help="run the reference-protocol pipeline (first sigma / window) and write the scene "
                    "file tests/host_cxx/house_replay.cpp replays")
    ap.add_argument("--sweep", action="store_true", help="sweep the simulator unknowns against the reference's two columns (profiles/round3_house_sweep.txt)")
    args = ap.parse_args()
    solve = make_solver(args.backend)
    if args.sweep:
        sweep(solve, args.frames)
        if args.backend == "oracle":
            function_tolerance_study(args.frames)
        sys.exit(0)
    if args.dump_scene:
        r = run_reference_protocol(float(args.sigmas.split(",")[0]), int(args.windows.split(",")[0]), solve, frames=args.frames, dump=args.dump_scene)
        r.pop("poses"); r.pop("window_digests")
        print(json.dumps(r))
        sys.exit(0)
    for s in [float(x) for x in args.sigmas.split(",")]:
        for W in [int(x) for x in args.windows.split(",")]:
            r = run(s, W, solve, frames=args.frames)
            r.pop("positions")
            ref = REFERENCE.get((s, W))
            if ref:
                r["reference"] = {"avg_iterations": ref[0], "avg_initial_cost": ref[1], "avg_final_cost": ref[2]}
            print(json.dumps(r), flush=True)
