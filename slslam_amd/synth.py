"""Deterministic synthetic inputs for the hot path.

The reference's datasets are not shipped (reference README:18-26), so tests and benches feed the
solver the same five arrays `SLAM::bundle_adjustment` builds (reference src/slam.cpp:899-920)
from a synthetic stereo line scene:

* camera model / constants: reference src/parameter.h:43-52 (640x480, f=406.05, cx=327.783,
  cy=237.172, baseline 0.12 along +x), pixel -> normalised as src/slam.cpp:121-128
* window shape: 2W keyframes, the W newest free, the rest fixed (src/slam.cpp:813-814, 855-870),
  newest keyframe exactly identity (metric_embedding, src/slam.cpp:1322)
* landmark initialisation: stereo triangulation of the first observation
  (SLAM::initialize_lm, src/slam.cpp:190-219) then gc_av_to_orth (src/gc.cpp:361-379)
* observations grouped by line in keyframe order, as the packer emits them (src/slam.cpp:848-882)

Pure numpy; no oracle, no GPU.
"""
import numpy as np

FOCAL = 406.05
CX = 327.783
CY = 237.172
WIDTH = 640
HEIGHT = 480
BASELINE = 0.12
HUBER_DELTA = 1.0 / FOCAL
INVERSE_DEPTH = 0.1


# ----------------------------------------------------------------------------- rotations
def rodrigues(w):
    """angle-axis -> rotation matrix (ceres::AngleAxisToRotationMatrix semantics, gc.cpp:24-36)."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * (K @ K)


def log_so3(R):
    """rotation matrix -> angle-axis (gc_Rodriguez(Matrix3d), gc.cpp:38-49)."""
    q = np.empty(4)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q[:] = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    sn = np.linalg.norm(q[1:])
    if sn < 1e-15:
        return 2.0 * q[1:]
    ang = 2.0 * (np.arctan2(-sn, -q[0]) if q[0] < 0 else np.arctan2(sn, q[0]))
    return q[1:] * (ang / sn)


def rt_to_wt(R, t):
    return np.concatenate([log_so3(R), t])


def wt_to_rt(wt):
    return rodrigues(wt[:3]), np.asarray(wt[3:6], dtype=np.float64)


def av_to_orth(av):
    """gc_av_to_orth, reference src/gc.cpp:361-379 (batched over leading dims)."""
    av = np.asarray(av, dtype=np.float64)
    a, v = av[..., :3], av[..., 3:]
    n = np.cross(a, v)
    nn = np.linalg.norm(n, axis=-1, keepdims=True)
    vn = np.linalg.norm(v, axis=-1, keepdims=True)
    x, y = n / nn, v / vn
    z = np.cross(x, y)
    o = np.empty(av.shape[:-1] + (4,))
    o[..., 0] = np.arctan2(y[..., 2], z[..., 2])
    o[..., 1] = np.arcsin(-x[..., 2])
    o[..., 2] = np.arctan2(x[..., 1], x[..., 0])
    o[..., 3] = np.arcsin(vn[..., 0] / np.sqrt(nn[..., 0] ** 2 + vn[..., 0] ** 2))
    return o


def orth_to_av(orth):
    """gc_orth_to_av, reference src/gc.cpp:419-442 (batched)."""
    orth = np.asarray(orth, dtype=np.float64)
    a, b, g, t = orth[..., 0], orth[..., 1], orth[..., 2], orth[..., 3]
    s1, c1, s2, c2, s3, c3 = np.sin(a), np.cos(a), np.sin(b), np.cos(b), np.sin(g), np.cos(g)
    d = np.cos(t) / np.sin(t)
    av = np.empty(orth.shape[:-1] + (6,))
    av[..., 0] = -(c1 * s2 * c3 + s1 * s3) * d
    av[..., 1] = -(c1 * s2 * s3 - s1 * c3) * d
    av[..., 2] = -(c1 * c2) * d
    av[..., 3] = s1 * s2 * c3 - c1 * s3
    av[..., 4] = s1 * s2 * s3 + c1 * c3
    av[..., 5] = s1 * c2
    return av


# ----------------------------------------------------------------------------- scene
def _trajectory(rng, num_kf):
    """Gently curving planar path with sinusoidal heave; returns camera-to-world (Rwc, c) per KF,
    oldest first.  Spacing ~0.75 m / <=15 deg per keyframe (reference src/parameter.h:59-60)."""
    yaw_rate = np.deg2rad(rng.uniform(-4.0, 4.0))
    phase = rng.uniform(0, 2 * np.pi)
    Rwc, c = [], []
    pos = np.zeros(3)
    yaw = 0.0
    for k in range(num_kf):
        Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        pitch = np.deg2rad(1.0) * np.sin(0.7 * k + phase)
        Rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
        R = Ry @ Rx
        p = pos.copy()
        p[1] += 0.05 * np.sin(0.9 * k + phase)
        Rwc.append(R)
        c.append(p)
        step = rng.uniform(0.6, 0.8)
        pos = pos + R[:, 2] * step
        yaw += yaw_rate + np.deg2rad(rng.normal(0, 0.5))
    return np.array(Rwc), np.array(c)


def _project(R, t, P):
    """world points P[...,3] -> (left px, right px, depth) for pose (R,t) world->camera."""
    Pc = P @ R.T + t
    z = Pc[..., 2]
    zl = np.where(np.abs(z) < 1e-9, 1e-9, z)
    ul = FOCAL * Pc[..., 0] / zl + CX
    vl = FOCAL * Pc[..., 1] / zl + CY
    ur = FOCAL * (Pc[..., 0] - BASELINE) / zl + CX
    return ul, vl, ur, z


def _initialize_lm(obs):
    """SLAM::initialize_lm (reference src/slam.cpp:190-219), batched over rows of obs[...,8]."""
    one = np.ones(obs.shape[:-1])
    p1 = np.stack([obs[..., 0], obs[..., 1], one], -1)
    p2 = np.stack([obs[..., 2], obs[..., 3], one], -1)
    p3 = np.stack([obs[..., 4] + BASELINE, obs[..., 5], one], -1)
    p4 = np.stack([obs[..., 6] + BASELINE, obs[..., 7], one], -1)
    o1 = np.zeros_like(p1)
    o2 = np.zeros_like(p1)
    o2[..., 0] = BASELINE

    def ppp(x1, x2, x3):  # gc_ppp_pi, gc.cpp:100-105
        n = np.cross(x1 - x3, x2 - x3)
        d = -np.sum(x3 * np.cross(x1, x2), -1)
        return np.concatenate([n, d[..., None]], -1)

    pi1, pi2 = ppp(p1, p2, o1), ppp(p3, p4, o2)
    dp = pi1[..., :, None] * pi2[..., None, :] - pi2[..., :, None] * pi1[..., None, :]
    n = np.stack([dp[..., 0, 3], dp[..., 1, 3], dp[..., 2, 3]], -1)            # gc_pipi_plk, gc.cpp:107-113
    v = np.stack([-dp[..., 1, 2], dp[..., 0, 2], -dp[..., 0, 1]], -1)
    cp = np.cross(v, n) / np.sum(v * v, -1, keepdims=True)                    # gc_plucker_origin
    cpn = np.linalg.norm(cp, axis=-1, keepdims=True)
    bad = (cpn < 0.1) | (cpn > 10.0)
    cp = np.where(bad, cp / cpn / INVERSE_DEPTH, cp)
    cp = np.where(cp[..., 2:3] < 0, -cp, cp)
    return np.concatenate([cp, v], -1)


def make_window(seed, num_lines=2000, num_kf=20, num_free=10, noise_px=0.5,
                pose_sigma_t=0.01, pose_sigma_r_deg=0.3, all_free=False,
                line_init="perturb", line_sigma_rel=0.01, line_sigma_dir_deg=0.5, mean_track=9.0):
    """One sliding-window LBA problem in the reference array contract.

    Returns a dict with the five arrays of src/slam.cpp:899-920 (`camera_index`, `line_index`,
    `fixed_index` [2M], `observations` [M,8], `parameters` [6C+4L]), sizes, and ground truth
    (`true_parameters`) for trajectory-error reporting.  Cameras 0..num_free-1 are the free
    keyframes (oldest..newest; the newest is exactly identity), the rest are the fixed ones.

    A line is tracked over a contiguous run of keyframes (mean length `mean_track`) inside its
    geometric visibility, which gives the M ~ 6 L of a real window.  `line_init`:
    "perturb" = landmarks already refined by earlier windows (truth + small error; the
    reference's windows start within a few % of their optimum cost);
    "triangulate" = every landmark freshly initialised from its first stereo observation
    (SLAM::initialize_lm) - the hard, far-from-optimum case.
    """
    if (num_kf if all_free else min(num_free, num_kf)) < 2:
        raise ValueError("a window line must be seen by >= 2 free keyframes (slam.cpp:839-840): num_free >= 2; "
                         "the 1-free-camera shape is make_motion_only()")
    rng = np.random.default_rng(np.random.SeedSequence([4, int(seed)]))   # rseed=4: main.cpp:26
    Rwc, cw = _trajectory(rng, num_kf)
    # re-root on the newest keyframe: world := newest camera frame
    Rn, cn = Rwc[-1], cw[-1]
    R_wc = np.einsum("ij,kjl->kil", Rn.T, Rwc)          # cam k -> new world
    c_w = (cw - cn) @ Rn
    R_cw = np.transpose(R_wc, (0, 2, 1))                # world -> camera (the reference's T.R)
    t_cw = -np.einsum("kij,kj->ki", R_cw, c_w)
    R_cw[-1] = np.eye(3)
    t_cw[-1] = 0.0
    if all_free:
        num_free = num_kf
    order = list(range(num_kf - num_free, num_kf)) + list(range(0, num_kf - num_free))  # cam idx -> kf
    kf_to_cam = np.empty(num_kf, dtype=np.int64)
    kf_to_cam[order] = np.arange(num_kf)

    # ---- lines: segments placed in the frustum union, kept if seen by >= 2 free keyframes
    ends_a, ends_b, vis_all = [], [], []
    have = 0
    while have < num_lines:
        n = max(256, 2 * (num_lines - have))
        kf = rng.integers(0, num_kf, n)
        u = rng.uniform(0, WIDTH, n)
        v = rng.uniform(0, HEIGHT, n)
        z = rng.uniform(2.0, 10.0, n)
        mid_c = np.stack([(u - CX) / FOCAL * z, (v - CY) / FOCAL * z, z], -1)
        mid_w = np.einsum("nij,nj->ni", R_wc[kf], mid_c) + c_w[kf]
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        half = 0.5 * rng.uniform(0.5, 3.0, n)[:, None]
        A, B = mid_w - d * half, mid_w + d * half
        vis = np.zeros((n, num_kf), dtype=bool)
        for k in range(num_kf):
            ok = np.ones(n, dtype=bool)
            for P in (A, B):
                ul, vl, ur, zz = _project(R_cw[k], t_cw[k], P)
                ok &= (zz > 0.5) & (ul >= 0) & (ul < WIDTH) & (ur >= 0) & (ur < WIDTH) & (vl >= 0) & (vl < HEIGHT)
            vis[:, k] = ok
        # tracker model: a contiguous run of keyframes around the seeding keyframe
        length = 2 + rng.poisson(max(mean_track - 2.0, 0.0), n)
        start = kf - rng.integers(0, length)
        kk = np.arange(num_kf)[None, :]
        vis &= (kk >= start[:, None]) & (kk < (start + length)[:, None])
        keep = vis[:, num_kf - num_free:].sum(1) >= 2                      # slam.cpp:839-840
        ends_a.append(A[keep]); ends_b.append(B[keep]); vis_all.append(vis[keep])
        have += int(keep.sum())
    A = np.concatenate(ends_a)[:num_lines]
    B = np.concatenate(ends_b)[:num_lines]
    vis = np.concatenate(vis_all)[:num_lines]

    # ---- noisy observations, grouped by line, keyframes in time order
    li, ki = np.nonzero(vis)                                               # row-major: by line, then kf
    M = len(li)
    obs = np.empty((M, 8))
    for k in range(num_kf):
        sel = ki == k
        if not sel.any():
            continue
        for e, P in enumerate((A, B)):
            ul, vl, ur, _ = _project(R_cw[k], t_cw[k], P[li[sel]])
            obs[sel, 2 * e] = ul
            obs[sel, 2 * e + 1] = vl
            obs[sel, 4 + 2 * e] = ur
            obs[sel, 4 + 2 * e + 1] = vl
    obs += rng.normal(0, noise_px, obs.shape) if noise_px > 0 else 0.0
    obs[:, 0::2] = (obs[:, 0::2] - CX) / FOCAL                             # slam.cpp:121-128
    obs[:, 1::2] = (obs[:, 1::2] - CY) / FOCAL

    # ---- parameters: true and perturbed poses
    true_cams = np.array([rt_to_wt(R_cw[k], t_cw[k]) for k in order])
    init_R, init_t = R_cw.copy(), t_cw.copy()
    for k in range(num_kf - num_free, num_kf - 1):                        # newest stays identity
        dw = rng.normal(0, np.deg2rad(pose_sigma_r_deg), 3)
        init_R[k] = rodrigues(dw) @ R_cw[k]
        init_t[k] = t_cw[k] + rng.normal(0, pose_sigma_t, 3)
    init_cams = np.array([rt_to_wt(init_R[k], init_t[k]) for k in order])

    # true lines: closest point + direction in the world frame
    dvec = (B - A) / np.linalg.norm(B - A, axis=1, keepdims=True)
    cp = A - np.sum(A * dvec, 1, keepdims=True) * dvec
    true_lines = av_to_orth(np.concatenate([cp, dvec], 1))
    # initial lines: triangulate the first observation in its keyframe, move to the world with
    # that keyframe's CURRENT (perturbed) pose: gc_line_from_pose(lm->line, init_kf->T), slam.cpp:884-886
    first = np.r_[True, li[1:] != li[:-1]]
    lm_c = _initialize_lm(obs[first])
    kf0 = ki[first]
    Rinv = np.transpose(init_R[kf0], (0, 2, 1))
    cp_w = np.einsum("nij,nj->ni", Rinv, lm_c[:, :3] - init_t[kf0])
    dv_w = np.einsum("nij,nj->ni", Rinv, lm_c[:, 3:])
    init_lines = av_to_orth(np.concatenate([cp_w, dv_w], 1))
    if line_init == "perturb":
        depth = np.linalg.norm(cp, axis=1, keepdims=True) + 1.0
        cp_p = cp + rng.normal(size=cp.shape) * line_sigma_rel * depth
        dw = rng.normal(0, np.deg2rad(line_sigma_dir_deg), cp.shape)
        dv_p = dvec + np.cross(dw, dvec)
        dv_p /= np.linalg.norm(dv_p, axis=1, keepdims=True)
        cp_p = cp_p - np.sum(cp_p * dv_p, 1, keepdims=True) * dv_p
        init_lines = av_to_orth(np.concatenate([cp_p, dv_p], 1))
    elif line_init != "triangulate":
        raise ValueError(line_init)

    cam_idx = kf_to_cam[ki].astype(np.int32)
    fixed = np.zeros((M, 2), dtype=np.int32)
    fixed[:, 0] = cam_idx >= num_free                                      # slam.cpp:855-870
    return {
        "num_cameras": num_kf, "num_lines": num_lines, "num_free_cameras": num_free,
        "camera_index": cam_idx, "line_index": li.astype(np.int32),
        "fixed_index": fixed.reshape(-1), "observations": obs,
        "parameters": np.concatenate([init_cams.reshape(-1), init_lines.reshape(-1)]),
        "true_parameters": np.concatenate([true_cams.reshape(-1), true_lines.reshape(-1)]),
        "baseline": BASELINE, "seed": int(seed),
    }


def make_motion_only(seed, num_lines=100, noise_px=0.5):
    """The motion_only_ba shape (reference src/slam.cpp:578-675): camera 0 = current estimate
    (free), camera 1 = identity (fixed), every line fixed, two observations per line."""
    w = make_window(seed, num_lines=num_lines, num_kf=2, num_free=2, noise_px=noise_px)
    M = len(w["camera_index"])
    # make_window orders cams oldest..newest with the newest = identity -> cam 1 is identity
    fixed = np.empty((M, 2), dtype=np.int32)
    fixed[:, 0] = w["camera_index"] == 1
    fixed[:, 1] = 1
    w["fixed_index"] = fixed.reshape(-1)
    w["num_free_cameras"] = 1
    return w


def make_pose_graph(seed, num_poses=260, num_loops=8, sigma_t=0.01, sigma_r_deg=0.2):
    """Closed-loop pose graph in the POProblem array contract (reference src/slam.cpp:1248-1280):
    sorted edge set (edge 0 = (0,1) fixes pose 0), constraints C = T_{n2<-n1}, parameters from
    dead-reckoned odometry (drift), loop-closure edges near the start/end of the loop."""
    rng = np.random.default_rng(np.random.SeedSequence([5, int(seed)]))
    N = num_poses
    radius = 0.75 * N / (2 * np.pi)
    Rs, ts = [], []
    for k in range(N):
        ang = 2 * np.pi * k / N
        c = np.array([radius * np.sin(ang), 0.05 * np.sin(5 * ang), radius * (1 - np.cos(ang))])
        yaw = ang
        Rwc = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        Rs.append(Rwc.T)
        ts.append(-Rwc.T @ c)
    # re-root on the last pose like metric_embedding (slam.cpp:1239-1240)? keep pose 0 = identity
    # frame: the reference re-roots on the newest kf; either way pose 0 is the gauge.
    edges = {}
    est_R, est_t = [Rs[0]], [ts[0]]
    for k in range(N - 1):
        Rrel = Rs[k + 1] @ Rs[k].T
        trel = ts[k + 1] - Rrel @ ts[k]
        Rn = rodrigues(rng.normal(0, np.deg2rad(sigma_r_deg), 3)) @ Rrel
        tn = trel + rng.normal(0, sigma_t, 3)
        edges[(k, k + 1)] = (Rn, tn)
        est_R.append(Rn @ est_R[k])
        est_t.append(Rn @ est_t[k] + tn)
    loops = 0
    cand = [(i, N - 1 - j) for i in range(0, 12) for j in range(0, 12)]
    rng.shuffle(cand)
    for (a, b) in cand:
        if loops >= num_loops:
            break
        if (a, b) in edges or a >= b:
            continue
        Rrel = Rs[b] @ Rs[a].T
        trel = ts[b] - Rrel @ ts[a]
        edges[(a, b)] = (rodrigues(rng.normal(0, np.deg2rad(0.05), 3)) @ Rrel, trel + rng.normal(0, 0.002, 3))
        loops += 1
    keys = sorted(edges.keys())                                            # std::set<pii> order
    p1 = np.array([k[0] for k in keys], dtype=np.int32)
    p2 = np.array([k[1] for k in keys], dtype=np.int32)
    cons = np.array([rt_to_wt(*edges[k]) for k in keys])
    params = np.array([rt_to_wt(est_R[k], est_t[k]) for k in range(N)])
    truth = np.array([rt_to_wt(Rs[k], ts[k]) for k in range(N)])
    return {"num_poses": N, "pose_index_1": p1, "pose_index_2": p2, "constraints": cons,
            "parameters": params.reshape(-1), "true_parameters": truth.reshape(-1), "seed": int(seed)}


def camera_centers(cams):
    """camera centres c = -R^T t of a [C,6] pose array (for trajectory-error reporting,
    reference matlab_script/calc_traj_err.m:28-40)."""
    cams = np.asarray(cams, dtype=np.float64).reshape(-1, 6)
    return np.array([-(rodrigues(c[:3]).T @ c[3:]) for c in cams])


def make_ransac_frame(seed, num_lines=150, num_hypotheses=256, noise_px=0.5):
    """Inputs of the RANSAC scoring loop (reference src/slam.cpp:396-413): the lines common to two
    frames in the previous keyframe's frame (world), their observations in the CURRENT frame, and a
    set of motion hypotheses around the true motion (some exact, some perturbed, some with |t| > 1
    which the reference skips).  Returns (poses [H,12] = R row-major | t, observations [K,8], lines [K,6],
    true_pose [12])."""
    rng = np.random.default_rng(np.random.SeedSequence([6, int(seed)]))
    w = make_window(seed, num_lines=num_lines, num_kf=2, num_free=2, noise_px=noise_px, mean_track=50.0)
    prm = w["true_parameters"]
    lines = orth_to_av(prm[12:].reshape(-1, 4))
    sel = w["camera_index"] == 0                      # camera 0 = the moving (current) frame, camera 1 = identity
    obs = np.zeros((num_lines, 8))
    has = np.zeros(num_lines, dtype=bool)
    obs[w["line_index"][sel]] = w["observations"][sel]
    has[w["line_index"][sel]] = True
    R, t = wt_to_rt(prm[:6])
    poses = []
    for h in range(num_hypotheses):
        s = [0.0, 1e-3, 1e-2, 0.1, 2.0][h % 5] * (1.0 if h < 5 else rng.uniform(0.2, 1.5))
        Rh = rodrigues(rng.normal(size=3) * 0.3 * s) @ R
        th = t + rng.normal(size=3) * s
        poses.append(np.concatenate([Rh.reshape(-1), th]))
    return np.array(poses), obs[has], lines[has], np.concatenate([R.reshape(-1), t])


def make_ransac_pair(seed, num_lines=150, noise_px=0.5, outlier_frac=0.2, num_trials=64, sample_size=5,
                     rot_deg=1.0, step_m=0.15):
    """Inputs of SLAM::ransac_motion (reference src/slam.cpp:322-427): the lines common to the previous
    and the current frame (world = previous frame), their stereo observations in both frames (a
    fraction of the current ones corrupted: wrong matches), and a pre-drawn sample sequence
    (rand.rand_sample draws sample_size distinct indices per trial).  Frame-to-frame motion is small
    (the generator linearises the rotation).  Returns dict(obs0, obs1, lines, samples, true_pose)."""
    rng = np.random.default_rng(np.random.SeedSequence([7, int(seed)]))
    # a dedicated small-motion pair: previous frame = identity, current frame = small motion
    wv = rng.normal(size=3) * np.deg2rad(rot_deg)
    tv = rng.normal(size=3) * step_m * np.array([0.3, 0.1, 1.0])
    R = rodrigues(wv)
    f, cx, cy, B = 406.05, 327.783, 237.172, 0.12
    obs0, obs1, lines = [], [], []
    while len(lines) < num_lines:
        c = np.array([rng.uniform(-4, 4), rng.uniform(-2, 2), rng.uniform(2.5, 10)])
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        h = rng.uniform(0.25, 1.5)
        P = np.stack([c - h * d, c + h * d])
        ok, rec = True, []
        for (Rc, tc) in ((np.eye(3), np.zeros(3)), (R, tv)):
            o = []
            for k in range(2):
                Q = (Rc @ P.T).T + tc - np.array([k * B, 0, 0])
                if np.any(Q[:, 2] < 0.5):
                    ok = False
                px = np.stack([f * Q[:, 0] / Q[:, 2] + cx, f * Q[:, 1] / Q[:, 2] + cy], axis=1)
                if np.any(px[:, 0] < 0) or np.any(px[:, 0] > 640) or np.any(px[:, 1] < 0) or np.any(px[:, 1] > 480):
                    ok = False
                px = px + rng.normal(size=(2, 2)) * noise_px
                o.append(((px - [cx, cy]) / f).reshape(-1))
            rec.append(np.concatenate(o))
        if not ok:
            continue
        dv = (P[1] - P[0]) / np.linalg.norm(P[1] - P[0])
        cp = P[0] - dv * (P[0] @ dv)
        obs0.append(rec[0]); obs1.append(rec[1]); lines.append(np.concatenate([cp, dv]))
    obs0, obs1, lines = np.array(obs0), np.array(obs1), np.array(lines)
    bad = rng.random(num_lines) < outlier_frac
    obs1[bad] += rng.normal(size=(int(bad.sum()), 8)) * 0.05
    samples = np.stack([rng.choice(num_lines, size=sample_size, replace=False) for _ in range(num_trials)]).astype(np.int32)
    return dict(obs0=obs0, obs1=obs1, lines=lines, samples=samples, true_pose=np.concatenate([R.reshape(-1), tv]),
                outliers=bad)
