"""Generates tests/golden/*.npz — run once in the build container, outputs committed.

The reference ships no golden vectors for this path and its solver (Ceres 1.7.0) is absent
(SURVEY.md 8c), so these fixtures pin the oracle with an INDEPENDENT second opinion:

* `residual_kat.npz`   residuals of reference src/lba_problem.h:46-118 and src/po_problem.h:68-108
                        from a numpy transcription written against the reference text with
                        scipy.spatial.transform.Rotation for all rotation algebra (no shared code
                        with oracle/*.c), plus central-difference Jacobians.
* `lba_optimum.npz`    a small window and the minimiser of sum_i rho(||r_i||^2)/2 found by
                        scipy.optimize.least_squares (trust-region reflective, numerical Jacobian)
                        on that transcription.
* `po_optimum.npz`     a small pose graph and its least-squares optimum, same method.
* `lba_lm_trace.npz`   iteration trace (cost, trust-region radius, gain ratio, accept / reject) of a Levenberg-
                        Marquardt loop written here in numpy from the Ceres 1.7.0 policy table on the
                        numpy transcription of the residual, central-difference Jacobians, dense normal equations:
                        a second, independent statement of the trust-region bookkeeping the oracle's lm_core.c restates.
* `traj_excerpt.txt`   DATA excerpt (25 lines) of two trajectory files the reference ships as results
                        (matlab_script/traj_slslam_itbt3f_basize10_wolc.txt lines 1-12 and
                        traj_slslam_myungdong_basize10_wlc.txt lines 100-112): output of the reference's
                        SLAM::save_trajectory (src/slam.cpp:1470-1496), used to pin the text format of
                        slslam_write_trajectory.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from slslam_amd import synth  # noqa: E402  (generator only: inputs, not answers)

B = 0.12
A_HUBER = 1.0 / 406.05


def line_residual_np(cam, line, obs):
    """numpy transcription of LineReprojectionError (reference src/lba_problem.h:46-118)."""
    a, b, g, t = line
    Rl = (Rotation.from_euler("z", g) * Rotation.from_euler("y", b) * Rotation.from_euler("x", a)).as_matrix()
    d = np.cos(t) / np.sin(t)
    cp = -Rl[:, 2] * d            # :66-68 is minus the third column times d
    dv = Rl[:, 1]                 # :70-72 is the second column
    Rc = Rotation.from_rotvec(cam[:3]).as_matrix()
    pc = Rc @ cp + cam[3:]
    dc = Rc @ dv
    out = []
    for k in range(2):
        p = pc - np.array([k * B, 0.0, 0.0])
        n = np.cross(p, dc)
        n = n / np.hypot(n[0], n[1])
        for e in range(2):
            x, y = obs[4 * k + 2 * e], obs[4 * k + 2 * e + 1]
            out.append(-(x * n[0] + y * n[1] + n[2]))
    return np.array(out)


def pose_residual_np(p1, p2, c):
    """numpy transcription of PoseConstraintError (reference src/po_problem.h:68-108):
    Te = T2^-1 * (C * T1), residual = (rotvec(Te), t(Te))."""
    def mat(p):
        return Rotation.from_rotvec(p[:3]).as_matrix(), np.asarray(p[3:])
    R1, t1 = mat(p1)
    R2, t2 = mat(p2)
    Rc, tc = mat(c)
    Rcc, tcc = Rc @ R1, Rc @ t1 + tc
    R2i, t2i = R2.T, -R2.T @ t2
    Re, te = R2i @ Rcc, R2i @ tcc + t2i
    return np.concatenate([Rotation.from_matrix(Re).as_rotvec(), te])


def central(f, x, h=1e-6):
    x = np.asarray(x, dtype=float)
    cols = []
    for k in range(len(x)):
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        cols.append((f(xp) - f(xm)) / (2 * h))
    return np.array(cols).T


def make_kat(rng):
    cams, lines, obss, res, jcs, jls = [], [], [], [], [], []
    for i in range(24):
        cam = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 1.0, 3)])
        if i == 0:
            cam[:3] = 0.0           # identity keyframe: first-order branch of AngleAxisRotatePoint
        line = np.array([rng.uniform(-3, 3), rng.uniform(-1.4, 1.4), rng.uniform(-3, 3), rng.uniform(0.1, 1.4)])
        obs = rng.uniform(-0.6, 0.6, 8)
        cams.append(cam); lines.append(line); obss.append(obs)
        res.append(line_residual_np(cam, line, obs))
        jcs.append(central(lambda c: line_residual_np(c, line, obs), cam))
        jls.append(central(lambda l: line_residual_np(cam, l, obs), line))
    p1s, p2s, cs, pres = [], [], [], []
    for i in range(16):
        p1 = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 2.0, 3)])
        p2 = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 2.0, 3)])
        c = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 1.0, 3)])
        p1s.append(p1); p2s.append(p2); cs.append(c); pres.append(pose_residual_np(p1, p2, c))
    np.savez(os.path.join(HERE, "residual_kat.npz"), cam=np.array(cams), line=np.array(lines), obs=np.array(obss),
             residual=np.array(res), j_cam_fd=np.array(jcs), j_line_fd=np.array(jls),
             pose1=np.array(p1s), pose2=np.array(p2s), constraint=np.array(cs), pose_residual=np.array(pres),
             survey_cam=np.array([0.01, -0.02, 0.03, 0.10, -0.20, 0.30]), survey_line=np.array([0.3, -0.4, 0.5, 0.6]),
             survey_obs=np.array([0.10, 0.05, -0.20, 0.15, 0.08, 0.05, -0.22, 0.15]),
             survey_residual=np.array([-0.6769197527221315, -0.4611016357636927, -0.5592511525774213, -0.350041545453428]),
             survey_residual_cam0=np.array([-0.5345227518574093, -0.31919053715071777, -0.44312099656355386, -0.23270941889772545]))


def make_lba(rng):
    w = synth.make_window(11, num_lines=24, num_kf=6, num_free=3, noise_px=0.3, mean_track=5.0)
    C, L = w["num_cameras"], w["num_lines"]
    cam_idx, line_idx, obs = w["camera_index"], w["line_index"], w["observations"]
    x0 = w["parameters"].copy()
    free_cam = np.array([c for c in range(C) if not w["fixed_index"].reshape(-1, 2)[cam_idx == c, 0].any()])
    idx = np.concatenate([np.concatenate([np.arange(6 * c, 6 * c + 6) for c in free_cam]), np.arange(6 * C, 6 * C + 4 * L)])

    def fun(z):
        x = x0.copy()
        x[idx] = z
        out = []
        for i in range(len(cam_idx)):
            r = line_residual_np(x[6 * cam_idx[i]:6 * cam_idx[i] + 6], x[6 * C + 4 * line_idx[i]:6 * C + 4 * line_idx[i] + 4], obs[i])
            s = r @ r
            rho = s if s <= A_HUBER ** 2 else 2 * A_HUBER * np.sqrt(s) - A_HUBER ** 2
            out.append(r * np.sqrt(rho / s) if s > 0 else r)     # ||out_i||^2 = rho(s_i)
        return np.concatenate(out)

    sol = least_squares(fun, x0[idx], method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=400, x_scale="jac")
    x = x0.copy()
    x[idx] = sol.x
    np.savez(os.path.join(HERE, "lba_optimum.npz"), num_cameras=C, num_lines=L, camera_index=cam_idx, line_index=line_idx,
             fixed_index=w["fixed_index"], observations=obs, parameters=x0, optimum=x, optimum_cost=0.5 * np.sum(sol.fun ** 2),
             initial_cost=0.5 * np.sum(fun(x0[idx]) ** 2), grad_inf=np.abs(sol.grad).max())
    print("lba optimum: cost %.6e -> %.6e, |grad|inf %.2e, nfev %d, status %d" % (
        0.5 * np.sum(fun(x0[idx]) ** 2), 0.5 * np.sum(sol.fun ** 2), np.abs(sol.grad).max(), sol.nfev, sol.status))


def make_po(rng):
    g = synth.make_pose_graph(3, num_poses=24, num_loops=3)
    N = g["num_poses"]
    x0 = g["parameters"].copy()

    def fun(z):
        x = np.concatenate([x0[:6], z])
        return np.concatenate([pose_residual_np(x[6 * a:6 * a + 6], x[6 * b:6 * b + 6], c)
                               for a, b, c in zip(g["pose_index_1"], g["pose_index_2"], g["constraints"])])

    sol = least_squares(fun, x0[6:], method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=200)
    x = np.concatenate([x0[:6], sol.x])
    np.savez(os.path.join(HERE, "po_optimum.npz"), num_poses=N, pose_index_1=g["pose_index_1"], pose_index_2=g["pose_index_2"],
             constraints=g["constraints"], parameters=x0, optimum=x, optimum_cost=0.5 * np.sum(sol.fun ** 2),
             initial_cost=0.5 * np.sum(fun(x0[6:]) ** 2))
    print("po optimum: cost %.6e -> %.6e, |grad|inf %.2e" % (0.5 * np.sum(fun(x0[6:]) ** 2), 0.5 * np.sum(sol.fun ** 2), np.abs(sol.grad).max()))


def make_lm_trace():
    w = synth.make_window(13, num_lines=20, num_kf=6, num_free=3, noise_px=0.6, mean_track=5.0)
    C, L = w["num_cameras"], w["num_lines"]
    cam_idx, line_idx, obs = w["camera_index"], w["line_index"], w["observations"]
    fi = w["fixed_index"].reshape(-1, 2)
    cam_const = np.array([fi[cam_idx == c, 0].any() for c in range(C)])
    free_cams = [c for c in range(C) if not cam_const[c] and (cam_idx == c).any()]
    cols = {("c", c): 6 * k for k, c in enumerate(free_cams)}
    for l in range(L):
        cols[("l", l)] = 6 * len(free_cams) + 4 * l
    n = 6 * len(free_cams) + 4 * L
    x = w["parameters"].copy()

    def blocks(xv):
        """robustified residuals and Jacobian (dense) + cost = sum rho / 2"""
        r_all, J, cost = np.zeros(4 * len(cam_idx)), np.zeros((4 * len(cam_idx), n)), 0.0
        for i, (c, l) in enumerate(zip(cam_idx, line_idx)):
            cam, ln = xv[6 * c:6 * c + 6], xv[6 * C + 4 * l:6 * C + 4 * l + 4]
            r = line_residual_np(cam, ln, obs[i])
            sq = r @ r
            if sq > A_HUBER ** 2:
                rho, rp = 2 * A_HUBER * np.sqrt(sq) - A_HUBER ** 2, A_HUBER / np.sqrt(sq)
            else:
                rho, rp = sq, 1.0
            cost += 0.5 * rho
            sr = np.sqrt(rp)                          # corrector for rho'' <= 0: scale r and J by sqrt(rho')
            r_all[4 * i:4 * i + 4] = sr * r
            if ("c", c) in cols:
                J[4 * i:4 * i + 4, cols[("c", c)]:cols[("c", c)] + 6] = sr * central(lambda q: line_residual_np(q, ln, obs[i]), cam)
            J[4 * i:4 * i + 4, cols[("l", l)]:cols[("l", l)] + 4] = sr * central(lambda q: line_residual_np(cam, q, obs[i]), ln)
        return r_all, J, cost

    def pack(xv):
        return np.concatenate([np.concatenate([xv[6 * c:6 * c + 6] for c in free_cams]), xv[6 * C:]])

    def unpack(xv, z):
        out = xv.copy()
        for k, c in enumerate(free_cams):
            out[6 * c:6 * c + 6] = z[6 * k:6 * k + 6]
        out[6 * C:] = z[6 * len(free_cams):]
        return out
    r, J, cost = blocks(x)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))                      # Jacobi scaling, once at x0
    g0 = np.abs(J.T @ r).max()
    radius, dec = 1e4, 2.0
    trace = [(0, cost, radius, 0.0, 0)]
    for it in range(1, 11):
        Js = J * scale
        g = Js.T @ r
        H = Js.T @ Js
        d2 = np.clip(np.diag(H), 1e-6, 1e32) / radius
        y = np.linalg.solve(H + np.diag(d2), g)
        delta = -scale * y
        model = 0.5 * y @ (g + d2 * y)
        z = pack(x)
        if np.linalg.norm(delta) <= 1e-8 * (np.linalg.norm(z) + 1e-8):
            break
        xc = unpack(x, z + delta)
        rc, Jc, costc = blocks(xc)
        change = cost - costc
        if abs(change) < 1e-6 * cost:
            break
        rho = change / model
        ok = rho > 1e-3
        if ok:
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), 1e16)
            dec = 2.0
            x, r, J, cost = xc, rc, Jc, costc
        else:
            radius /= dec
            dec *= 2.0
        trace.append((it, cost, radius, rho, int(ok)))
        if ok and np.abs(J.T @ r).max() <= 1e-10 * max(g0, 1e-12):
            break
    tr = np.array(trace)
    np.savez(os.path.join(HERE, "lba_lm_trace.npz"), num_cameras=C, num_lines=L, camera_index=cam_idx, line_index=line_idx,
             fixed_index=w["fixed_index"], observations=obs, parameters=w["parameters"], iteration=tr[:, 0].astype(int), cost=tr[:, 1],
             radius=tr[:, 2], relative_decrease=tr[:, 3], successful=tr[:, 4].astype(int), final_parameters=x)
    print("lm trace: %d records, cost %.6e -> %.6e, %d rejected" % (len(tr), tr[0, 1], tr[-1, 1], int((tr[1:, 4] == 0).sum())))


def make_traj_excerpt():
    ref = "/root/reference/matlab_script"
    a = open(os.path.join(ref, "traj_slslam_itbt3f_basize10_wolc.txt")).read().splitlines(True)[:12]
    b = open(os.path.join(ref, "traj_slslam_myungdong_basize10_wlc.txt")).read().splitlines(True)[99:112]
    open(os.path.join(HERE, "traj_excerpt.txt"), "w").write("".join(a + b))


if __name__ == "__main__":
    make_traj_excerpt()
    make_lm_trace()
    rng = np.random.default_rng(20260927)
    make_kat(rng)
    make_lba(rng)
    make_po(rng)
