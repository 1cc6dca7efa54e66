"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
and only as the checker / timed CPU baseline.  The product package (slslam_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
if os.environ.get("SLSLAM_ORACLE_LIB"):          # tests/test_sanitizers.py: the same sources built with -fsanitize=address,undefined
    _LIB_PATH = os.environ["SLSLAM_ORACLE_LIB"]


def build(force=False):
    """Compile the C restatement with gcc (building the checker is not using it)."""
    if os.environ.get("SLSLAM_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("lm_core.c", "lba_oracle.c", "po_oracle.c", "ransac_oracle.c", "jet_impl.h", "lm_core.h", "slslam_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class LMOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double),
                ("max_num_consecutive_invalid_steps", C.c_int),
                ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int),
                ("linear_solver", C.c_int),
                ("policy_variant", C.c_int)]


class Summary(C.Structure):
    _fields_ = [("num_successful_steps", C.c_int),
                ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double),
                ("final_cost", C.c_double),
                ("fixed_cost", C.c_double),
                ("termination_type", C.c_int),
                ("num_free_parameters", C.c_int),
                ("num_residual_blocks", C.c_int)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double)]


class LBAProblem(C.Structure):
    _fields_ = [("num_cameras", C.c_int), ("num_lines", C.c_int), ("num_observations", C.c_int),
                ("camera_index", C.POINTER(C.c_int)), ("line_index", C.POINTER(C.c_int)),
                ("fixed_index", C.POINTER(C.c_int)), ("observations", C.POINTER(C.c_double)),
                ("baseline", C.c_double), ("huber_delta", C.c_double)]


class POProblem(C.Structure):
    _fields_ = [("num_poses", C.c_int), ("num_edges", C.c_int),
                ("pose_index_1", C.POINTER(C.c_int)), ("pose_index_2", C.POINTER(C.c_int)),
                ("constraints", C.POINTER(C.c_double))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _lib.oracle_lm_default_options.argtypes = [C.POINTER(LMOptions)]
        _lib.oracle_line_residual.argtypes = [dp, dp, dp, C.c_double, dp]
        _lib.oracle_line_residual_jet.argtypes = [dp, dp, dp, C.c_double, dp, dp, dp]
        _lib.oracle_huber.argtypes = [C.c_double, C.c_double, dp]
        _lib.oracle_lba_cost.argtypes = [C.POINTER(LBAProblem), dp, dp, dp, dp]
        _lib.oracle_lba_cost.restype = C.c_double
        _lib.oracle_lba_solve.argtypes = [C.POINTER(LBAProblem), C.POINTER(LMOptions), dp,
                                          C.POINTER(Summary), C.POINTER(Iteration), C.c_int, ip]
        _lib.oracle_lba_solve_many.argtypes = [C.c_int, C.POINTER(LBAProblem), C.POINTER(LMOptions), C.POINTER(dp),
                                               C.POINTER(Summary), C.c_int]
        _lib.oracle_pose_residual.argtypes = [dp, dp, dp, dp]
        _lib.oracle_pose_residual_jet.argtypes = [dp, dp, dp, dp, dp, dp]
        _lib.oracle_po_cost.argtypes = [C.POINTER(POProblem), dp]
        _lib.oracle_po_cost.restype = C.c_double
        _lib.oracle_po_solve.argtypes = [C.POINTER(POProblem), C.POINTER(LMOptions), dp,
                                         C.POINTER(Summary), C.POINTER(Iteration), C.c_int, ip]
        _lib.oracle_ransac_score.argtypes = [C.c_int, dp, C.c_int, dp, dp, C.c_double, C.c_double, ip, C.POINTER(C.c_ubyte)]
        _lib.oracle_vo_angle_axis_approx.argtypes = [C.c_int, dp, dp, C.c_double, dp]
        _lib.oracle_ransac_motion.argtypes = [C.c_int, dp, dp, dp, C.c_int, C.c_int, ip, C.c_double, C.c_double, C.c_double,
                                              C.c_int, ip, dp, C.POINTER(C.c_ubyte)]
        _lib.oracle_av_to_orth.argtypes = [dp, dp]
        _lib.oracle_orth_to_av.argtypes = [dp, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def default_options(**kw):
    o = LMOptions()
    lib().oracle_lm_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def line_residual(camera, line, obs, baseline=0.12):
    camera, line, obs = _f64(camera), _f64(line), _f64(obs)
    r = np.zeros(4)
    lib().oracle_line_residual(_dp(camera), _dp(line), _dp(obs), baseline, _dp(r))
    return r


def line_residual_jet(camera, line, obs, baseline=0.12):
    camera, line, obs = _f64(camera), _f64(line), _f64(obs)
    r, jc, jl = np.zeros(4), np.zeros((4, 6)), np.zeros((4, 4))
    lib().oracle_line_residual_jet(_dp(camera), _dp(line), _dp(obs), baseline, _dp(r), _dp(jc), _dp(jl))
    return r, jc, jl


def huber(s, a):
    rho = np.zeros(3)
    lib().oracle_huber(float(s), float(a), _dp(rho))
    return rho


def _summary_dict(s):
    return {k: getattr(s, k) for k, _ in Summary._fields_}


def _trace_list(tr, n):
    return [{k: getattr(tr[i], k) for k, _ in Iteration._fields_} for i in range(n)]


class _LBAHandle:
    def __init__(self, w, huber_delta):
        self.cam = _i32(w["camera_index"])
        self.line = _i32(w["line_index"])
        self.fixed = _i32(w["fixed_index"])
        self.obs = _f64(w["observations"]).reshape(-1)
        self.p = LBAProblem(int(w["num_cameras"]), int(w["num_lines"]), int(len(self.cam)),
                            _ip(self.cam), _ip(self.line), _ip(self.fixed), _dp(self.obs),
                            float(w.get("baseline", 0.12)), float(huber_delta))


def lba_cost(w, params, huber_delta=1.0 / 406.05, want_jac=False):
    h = _LBAHandle(w, huber_delta)
    params = _f64(params)
    m = h.p.num_observations
    if want_jac:
        r, jc, jl = np.zeros((m, 4)), np.zeros((m, 4, 6)), np.zeros((m, 4, 4))
        c = lib().oracle_lba_cost(C.byref(h.p), _dp(params), _dp(r), _dp(jc), _dp(jl))
        return c, r, jc, jl
    return lib().oracle_lba_cost(C.byref(h.p), _dp(params), None, None, None)


def lba_solve(w, params=None, huber_delta=1.0 / 406.05, trace_cap=256, **opt):
    """Solve one window. w: dict with num_cameras, num_lines, camera_index, line_index,
    fixed_index, observations, (parameters). Returns (params_out, summary dict, trace list)."""
    h = _LBAHandle(w, huber_delta)
    x = _f64(w["parameters"] if params is None else params).copy()
    o = default_options(**opt)
    s = Summary()
    tr = (Iteration * trace_cap)()
    n = C.c_int(0)
    rc = lib().oracle_lba_solve(C.byref(h.p), C.byref(o), _dp(x), C.byref(s), tr, trace_cap, C.byref(n))
    d = _summary_dict(s)
    d["rc"] = rc
    return x, d, _trace_list(tr, min(n.value, trace_cap))


def lba_solve_many(windows, num_threads, huber_delta=1.0 / 406.05, **opt):
    """Independent windows fanned out over host threads inside the C library (OpenMP).
    Returns (list of solved parameter vectors, list of summary dicts)."""
    hs = [_LBAHandle(w, huber_delta) for w in windows]
    xs = [_f64(w["parameters"]).copy() for w in windows]
    probs = (LBAProblem * len(hs))(*[h.p for h in hs])
    ptrs = (C.POINTER(C.c_double) * len(xs))(*[_dp(x) for x in xs])
    sums = (Summary * len(hs))()
    o = default_options(**opt)
    lib().oracle_lba_solve_many(len(hs), probs, C.byref(o), ptrs, sums, int(num_threads))
    return xs, [_summary_dict(s) for s in sums]


def pose_residual_jet(p1, p2, c):
    p1, p2, c = _f64(p1), _f64(p2), _f64(c)
    r, j1, j2 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
    lib().oracle_pose_residual_jet(_dp(p1), _dp(p2), _dp(c), _dp(r), _dp(j1), _dp(j2))
    return r, j1, j2


class _POHandle:
    def __init__(self, g):
        self.i1 = _i32(g["pose_index_1"])
        self.i2 = _i32(g["pose_index_2"])
        self.c = _f64(g["constraints"]).reshape(-1)
        self.p = POProblem(int(g["num_poses"]), int(len(self.i1)), _ip(self.i1), _ip(self.i2), _dp(self.c))


def po_cost(g, params):
    h = _POHandle(g)
    params = _f64(params)
    return lib().oracle_po_cost(C.byref(h.p), _dp(params))


def po_solve(g, params=None, trace_cap=256, **opt):
    h = _POHandle(g)
    x = _f64(g["parameters"] if params is None else params).copy()
    o = default_options(**opt)
    s = Summary()
    tr = (Iteration * trace_cap)()
    n = C.c_int(0)
    rc = lib().oracle_po_solve(C.byref(h.p), C.byref(o), _dp(x), C.byref(s), tr, trace_cap, C.byref(n))
    d = _summary_dict(s)
    d["rc"] = rc
    return x, d, _trace_list(tr, min(n.value, trace_cap))


def av_to_orth(av):
    av = _f64(av)
    o = np.zeros(4)
    lib().oracle_av_to_orth(_dp(av), _dp(o))
    return o


def orth_to_av(orth):
    orth = _f64(orth)
    o = np.zeros(6)
    lib().oracle_orth_to_av(_dp(orth), _dp(o))
    return o


def ransac_score(poses, observations, lines, baseline=0.12, error_thr=5.0 / 406.05):
    poses = _f64(poses).reshape(-1, 12)
    obs, ln = _f64(observations).reshape(-1, 8), _f64(lines).reshape(-1, 6)
    h, k = len(poses), len(obs)
    scores = np.zeros(max(h, 1), dtype=np.int32)
    inl = np.zeros(max(h * k, 1), dtype=np.uint8)
    lib().oracle_ransac_score(h, _dp(poses), k, _dp(obs), _dp(ln), float(baseline), float(error_thr), _ip(scores),
                              inl.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return scores[:h], inl[:h * k].reshape(h, k).astype(bool)


def vo_angle_axis_approx(obs0, obs1, baseline=-0.12):
    """SLAM::vo_angle_axis_approx on s sampled correspondences -> (num_solutions, pose[12])."""
    o0, o1 = _f64(obs0).reshape(-1, 8), _f64(obs1).reshape(-1, 8)
    pose = np.zeros(12)
    n = lib().oracle_vo_angle_axis_approx(len(o0), _dp(o0), _dp(o1), float(baseline), _dp(pose))
    return n, pose


def ransac_motion(obs0, obs1, lines, samples, baseline=0.12, error_thr=5.0 / 406.05, prob_free_outliers=0.999,
                  max_trials=1000, best_score=0):
    """SLAM::ransac_motion over a given sample sequence [T, s] -> (trial_cnt, best_score, best_pose, inliers)."""
    o0, o1, ln = _f64(obs0).reshape(-1, 8), _f64(obs1).reshape(-1, 8), _f64(lines).reshape(-1, 6)
    smp = np.ascontiguousarray(samples, dtype=np.int32)
    k = len(o0)
    bs = np.array([best_score], dtype=np.int32)
    pose = np.zeros(12)
    inl = np.zeros(max(k, 1), dtype=np.uint8)
    tc = lib().oracle_ransac_motion(k, _dp(o0), _dp(o1), _dp(ln), smp.shape[1], smp.shape[0], _ip(smp), float(baseline),
                                    float(error_thr), float(prob_free_outliers), int(max_trials), _ip(bs), _dp(pose),
                                    inl.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return tc, int(bs[0]), pose, inl[:k].astype(bool)
