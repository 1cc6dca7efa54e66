"""GPU parity tests of the line bundle adjustment: the HIP path (through the C ABI) against the
oracle on the same seeded inputs, against the committed golden optimum, and - at the bench's full
window size - through size-independent properties.  Everything here needs a real MI355X."""
import os

import numpy as np
import pytest

from slslam_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp64 on both sides.  The initial evaluation and the first step agree to round-off (different
# summation order: wave shuffles / LDS atomics vs serial loops, FMA contraction).  Later iterations
# inherit the conditioning of the problem: the oracle's own dense and Schur back-ends drift apart
# at the same rate (tests/test_oracle.py::test_dense_and_schur_linear_solvers_agree), so the
# tolerance widens with the iteration index.  Stated tolerances: cost rel 1e-13 (it 0),
# 1e-10 (it 1), 1e-6 (later); final parameters 1e-5; final cost rel 1e-7.
#
# MEASURED (tools/parity_study.py on the MI355X, profiles/round2_parity_study.txt; the eleven shapes of
# test_solve_trace_matches_oracle, test_long_and_short_line_runs and the 2000-line bench window): over ALL iterations
# cost <= 1.1e-8, radius <= 2.8e-7, step norm <= 4.1e-8, gain ratio <= 1.9e-6 (relative); final camera parameters
# <= 7.4e-11, final line parameters <= 4.5e-8 with a median of 3.5e-11 - the tail is a handful of depth-degenerate lines
# per window (seen under a few degrees of parallax: an eigenvalue of their 4x4 block ~1e-8 of the others, the lines
# profiles/round2_rejection_study.txt names), and the oracle's own dense-vs-Schur runs differ MORE on the same lines
# (1.1e-7).  TIGHT below = those figures with a margin of ~30: asserted on every iteration of the studied shapes.
REL = 1e-9
REL_BY_ITER = {0: 1e-13, 1: 1e-10}
TIGHT = dict(cost=3e-7, radius=1e-5, step_norm=2e-6, rho=5e-5, cam=3e-9, line=2e-6)


def _assert_trace_parity(t_ref, t_hip, n=None, tight=False):
    assert len(t_ref) == len(t_hip)
    for a, b in list(zip(t_ref, t_hip))[:n]:
        assert a["iteration"] == b["iteration"]
        assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"]
        assert abs(a["cost"] - b["cost"]) <= REL_BY_ITER.get(a["iteration"], 1e-6) * abs(a["cost"])
        assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-6 * a["trust_region_radius"]
        assert abs(a["step_norm"] - b["step_norm"]) <= 1e-5 * (a["step_norm"] + 1e-12)
        assert abs(a["relative_decrease"] - b["relative_decrease"]) <= 1e-4 * (abs(a["relative_decrease"]) + 1e-3)
    if tight:                                         # every iteration, measured tolerances
        for a, b in zip(t_ref, t_hip):
            assert a["step_is_successful"] == b["step_is_successful"] and a["step_is_valid"] == b["step_is_valid"]
            assert abs(a["cost"] - b["cost"]) <= TIGHT["cost"] * abs(a["cost"])
            assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= TIGHT["radius"] * a["trust_region_radius"]
            assert abs(a["step_norm"] - b["step_norm"]) <= TIGHT["step_norm"] * (a["step_norm"] + 1e-12)
            assert abs(a["relative_decrease"] - b["relative_decrease"]) <= TIGHT["rho"] * (abs(a["relative_decrease"]) + 1e-3)


def _assert_params_parity(w, x_ref, x_hip):
    """Final parameters by class: camera poses to 3e-9, line parameters to 2e-6 (see the measured figures above)."""
    nc = 6 * int(w["num_cameras"])
    assert np.abs(x_ref[:nc] - x_hip[:nc]).max() <= TIGHT["cam"]
    assert np.abs(x_ref[nc:] - x_hip[nc:]).max() <= TIGHT["line"]


def _assert_summary_parity(s_ref, s_hip):
    for k in ("num_successful_steps", "num_unsuccessful_steps", "termination_type", "num_free_parameters", "num_residual_blocks"):
        assert s_ref[k] == s_hip[k], k
    for k, tol in (("initial_cost", 1e-12), ("fixed_cost", 1e-12), ("final_cost", 1e-7)):
        assert abs(s_ref[k] - s_hip[k]) <= tol * abs(s_ref[k]) + 1e-300, k


def test_linearise_matches_oracle(hip, oracle):
    """Residuals + analytic Jacobians of the kernels vs the dual-number restatement of
    AutoDiffCostFunction<LineReprojectionError,4,6,4> + HuberLoss corrector."""
    for seed, robust in ((1, True), (2, False)):
        w = synth.make_window(seed, num_lines=150)
        b = hip.LBABatch()
        b.add(w)
        b.finalize(huber_delta=(1.0 / 406.05 if robust else 0.0))
        c, r, jc, jl = b.linearise(0, len(w["camera_index"]))
        c0, r0, jc0, jl0 = oracle.lba_cost(w, w["parameters"], huber_delta=(1.0 / 406.05 if robust else 0.0), want_jac=True)
        assert abs(c - c0) <= 1e-13 * c0
        assert np.abs(r - r0).max() < 1e-14
        assert (np.abs(jc - jc0) / (1 + np.abs(jc0))).max() < 1e-12
        assert (np.abs(jl - jl0) / (1 + np.abs(jl0))).max() < 1e-12
        b.close()


@pytest.mark.parametrize("seed,lines,kf,free", [(1, 60, 20, 10), (2, 200, 20, 10), (3, 500, 20, 10), (4, 80, 6, 3), (5, 40, 20, 20),
                                                 (6, 150, 40, 20), (7, 150, 80, 40)])     # W = 20 and W = 40 of the reference's study
def test_solve_trace_matches_oracle(hip, oracle, seed, lines, kf, free):
    """LM iteration by iteration: cost, gain ratio, radius, step norm, accept/reject decisions."""
    w = synth.make_window(seed, num_lines=lines, num_kf=kf, num_free=free)
    x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
    x1, s1, t1 = hip.lba_solve(w)
    _assert_trace_parity(t0, t1, n=4, tight=True)      # the first iterations agree to round-off ...
    _assert_summary_parity(s0, s1)
    # ... later ones inherit the conditioning of the problem (oracle dense-vs-Schur differ as much)
    _assert_params_parity(w, x0, x1)
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-7 * s0["final_cost"]


def test_long_and_short_line_runs(hip, oracle):
    """Lane layout corner cases of the packer: lines observed by more than 16 keyframes (their run of lanes spans
    several 16-lane rows of a tile) and lines with 1-3 observations (several sin/cos rounds per lane in the
    back-substitution), alone and mixed with ordinary lines in one window."""
    long_w = synth.make_window(11, num_lines=60, num_kf=24, num_free=12, mean_track=40.0)
    counts = np.bincount(long_w["line_index"], minlength=long_w["num_lines"])
    assert counts.max() > 16
    short_w = synth.make_window(12, num_lines=90, num_kf=8, num_free=6, mean_track=2.0)
    # thin some tracks of the mixed window down to a single observation (the line stays in the problem)
    mixed = synth.make_window(13, num_lines=120, num_kf=24, num_free=12, mean_track=12.0)
    keep = np.ones(len(mixed["camera_index"]), dtype=bool)
    for l in range(0, 120, 7):
        idx = np.nonzero(mixed["line_index"] == l)[0]
        keep[idx[1:]] = False
    mixed = dict(mixed, camera_index=mixed["camera_index"][keep], line_index=mixed["line_index"][keep],
                 observations=mixed["observations"][keep], fixed_index=mixed["fixed_index"].reshape(-1, 2)[keep].reshape(-1))
    cm = np.bincount(mixed["line_index"], minlength=120)
    assert cm.min() == 1 and cm.max() > 16
    # the largest shape the path takes: 64 keyframes, lines observed by all of them (runs of 64 lanes = a whole tile)
    full_w = synth.make_window(21, num_lines=40, num_kf=64, num_free=12, mean_track=300.0)
    assert np.bincount(full_w["line_index"]).max() == 64
    for w in (long_w, short_w, mixed, full_w):
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = hip.lba_solve(w)
        _assert_trace_parity(t0, t1, n=3, tight=True)
        _assert_summary_parity(s0, s1)
        _assert_params_parity(w, x0, x1)


def test_windows_beyond_the_tiled_sweeps(hip, oracle):
    """The reference's window size is a flag (src/main.cpp:22) and its study runs W = 40: 80 keyframes, 40 free, lines tracked
    through all of them.  Such windows (more than 20 free / 64 cameras, or a line with more than 64 observations) take the
    global-memory path (lba_big.h: per-observation Jacobians in HBM, every sum a gather in list order, reduced system on the
    pose-graph MFMA Cholesky); same algorithm, so the same traces, and reproducible bit for bit.
    (Lines observed by more than 64 keyframes: tests/test_house_study.py, W = 40.)"""
    ws = [synth.make_window(31, num_lines=100, num_kf=80, num_free=40, mean_track=30.0),
          synth.make_window(33, num_lines=80, num_kf=70, num_free=12, mean_track=20.0),        # > 64 cameras only
          synth.make_window(34, num_lines=50, num_kf=30, num_free=24, mean_track=10.0),        # > 20 free cameras only
          synth.make_window(36, num_lines=60, num_kf=50, num_free=42, mean_track=25.0),        # 252 unknowns: the 17-slot form of k_big_solve
          synth.make_window(37, num_lines=60, num_kf=50, num_free=44, mean_track=25.0)]        # 264 unknowns: the launch chain of the pose-graph Cholesky
    for w in ws:
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = hip.lba_solve(w)
        _assert_trace_parity(t0, t1, n=4, tight=True)       # every iteration at the tolerances of the tiled path
        _assert_summary_parity(s0, s1)
        _assert_params_parity(w, x0, x1)
        x2, s2, t2 = hip.lba_solve(w)                       # gather sums in list order: bitwise reproducible (round 2: atomics)
        assert np.array_equal(x1, x2) and s1 == s2 and t1 == t2
    # a batch that mixes oversize windows with ordinary ones is solved as two batches side by side - the ordinary windows keep the
    # tiled sweeps - and says so; every window's result is what it is in a batch of its own, bit for bit
    b = hip.LBABatch()
    mix = [ws[0], synth.make_window(35, num_lines=120), synth.make_motion_only(6, num_lines=30), ws[2], synth.make_window(38, num_lines=300)]
    for w in mix:
        b.add(w)
    b.finalize(); b.solve(); b.download()
    assert b.path() == 3                                     # SLSLAM_PATH_MIXED
    first = [b.parameters(i).copy() for i in range(len(mix))]
    # (device buffer through the HIP runtime the library itself is linked to)
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    nbytes = 8 * b.total_parameters()
    dptr = ctypes.c_void_p()
    assert rt.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(nbytes)) == 0
    b.export_device(dptr.value)
    flat = np.zeros(b.total_parameters())
    assert rt.hipDeviceSynchronize() == 0
    assert rt.hipMemcpy(flat.ctypes.data_as(ctypes.c_void_p), dptr, ctypes.c_size_t(nbytes), 2) == 0      # hipMemcpyDeviceToHost
    assert rt.hipFree(dptr) == 0
    off = 0
    for i, w in enumerate(mix):
        n = len(w["parameters"])
        assert np.array_equal(flat[off:off + n], first[i])       # the device export keeps the caller's order
        off += n
        alone = hip.LBABatch(); alone.add(w); alone.finalize(chunks_per_window=b.window_chunks(i)); alone.solve(); alone.download()
        if alone.path() == 1:                                   # a motion-only problem alone takes its one-launch kernel: other sums
            assert np.abs(alone.parameters(0) - first[i]).max() < 1e-9, i
        else:
            assert np.array_equal(alone.parameters(0), first[i]), i
            assert alone.summary(0) == b.summary(i) and alone.trace(0) == b.trace(i), i
        alone.close()
    assert b.counts()["windows"] == len(mix)
    for i, w in enumerate(mix):
        xo, so, to = oracle.lba_solve(w, linear_solver=1)
        _assert_params_parity(w, xo, first[i])
        _assert_summary_parity(so, b.summary(i))
        _assert_trace_parity(to, b.trace(i), n=4, tight=True)
    b.reset(); b.solve(); b.download()
    for i in range(len(mix)):
        assert np.array_equal(first[i], b.parameters(i))
    b.close()
    b = hip.LBABatch()
    b.add(mix[1]); b.finalize()
    assert b.path() == 0                                     # SLSLAM_PATH_TILED
    b.close()
    b = hip.LBABatch()
    b.add(mix[2]); b.finalize()
    assert b.path() == 1                                     # SLSLAM_PATH_FUSED_MOTION_ONLY
    b.close()
    # max_num_iterations = 0: the initial evaluation only
    x1, s1, t1 = hip.lba_solve(ws[0], max_num_iterations=0)
    x0, s0, t0 = oracle.lba_solve(ws[0], linear_solver=1, max_num_iterations=0)
    assert abs(s1["initial_cost"] - s0["initial_cost"]) <= 1e-12 * s0["initial_cost"] and np.array_equal(x1, ws[0]["parameters"])


def test_oversize_window_sizes_and_batches(hip, oracle):
    """The one-launch reduced solve of the global-memory path (lba_big_solve.h: blocks of 16 in registers, look-ahead, updates
    deferred behind a double-buffered panel) over system sizes that end inside a block, on a block edge, in the 15- and the
    17-slot form; and a batch of oversize windows of different sizes solved twice (graph replay)."""
    for seed, free, kf, lines, mt in ((101, 21, 42, 40, 20.0), (103, 30, 66, 50, 28.0), (104, 33, 80, 74, 50.0), (105, 37, 74, 90, 33.0),
                                      (106, 40, 85, 74, 61.0), (107, 41, 82, 64, 40.0), (109, 22, 30, 35, 9.0)):
        w = synth.make_window(seed, num_lines=lines, num_kf=kf, num_free=free, mean_track=mt)
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = hip.lba_solve(w)
        _assert_trace_parity(t0, t1, n=4, tight=True)
        _assert_summary_parity(s0, s1)
        _assert_params_parity(w, x0, x1)
    ws = [synth.make_window(200 + i, num_lines=40 + 7 * i, num_kf=50 + 5 * i, num_free=24 + 3 * i, mean_track=20.0 + i) for i in range(6)]
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize(); b.solve(); b.download()
    assert b.path() == 2
    first = [b.parameters(i).copy() for i in range(len(ws))]
    b.reset(); b.solve(); b.download()
    for i, w in enumerate(ws):
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        assert np.array_equal(first[i], b.parameters(i))
        _assert_params_parity(w, x0, first[i])
        _assert_summary_parity(s0, b.summary(i))
        _assert_trace_parity(t0, b.trace(i), n=4, tight=True)
    b.close()


def test_every_camera_constant(hip, oracle):
    """Lines free, every camera constant (fixed_index, lba_problem.cpp:81-90): the reduced camera system is empty, every line is
    solved on its own."""
    w = synth.make_window(11, num_lines=60)
    fi = np.asarray(w["fixed_index"]).reshape(-1, 2).copy()
    fi[:, 0] = 1
    w = dict(w); w["fixed_index"] = fi.reshape(-1).astype(np.int32)
    x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
    x1, s1, t1 = hip.lba_solve(w)
    _assert_trace_parity(t0, t1, n=3)
    _assert_summary_parity(s0, s1)
    _assert_params_parity(w, x0, x1)
    ncam = int(w["num_cameras"])
    assert np.array_equal(x1[:6 * ncam], np.asarray(w["parameters"])[:6 * ncam])


def test_one_iteration_is_roundoff_exact(hip, oracle):
    w = synth.make_window(11, num_lines=300)
    x0, s0, t0 = oracle.lba_solve(w, linear_solver=0, max_num_iterations=1)      # dense normal equations
    x1, s1, t1 = hip.lba_solve(w, max_num_iterations=1)
    _assert_trace_parity(t0, t1)
    assert np.abs(x0 - x1).max() < 1e-9


def test_motion_only_shape(hip, oracle):
    """motion_only_ba (slam.cpp:578-675): one free camera, identity camera + every line constant."""
    w = synth.make_motion_only(2, num_lines=60)
    x0, s0, t0 = oracle.lba_solve(w)
    x1, s1, t1 = hip.lba_solve(w)
    _assert_trace_parity(t0, t1)
    _assert_summary_parity(s0, s1)
    assert s1["fixed_cost"] > 0 and s1["num_free_parameters"] == 6
    assert np.array_equal(x1[6:], w["parameters"][6:])
    assert np.abs(x0[:6] - x1[:6]).max() < 1e-10


def test_non_robust_and_unscaled_variants(hip, oracle):
    w = synth.make_window(6, num_lines=100)
    for kw_h, kw_o in (({"huber_delta": 0.0}, {"huber_delta": 0.0}), ({"jacobi_scaling": 0}, {"jacobi_scaling": 0})):
        o_args = {k: v for k, v in kw_o.items() if k != "huber_delta"}
        x0, s0, t0 = oracle.lba_solve(w, huber_delta=kw_o.get("huber_delta", 1.0 / 406.05), linear_solver=1, **o_args)
        x1, s1, t1 = hip.lba_solve(w, **kw_h)
        _assert_trace_parity(t0, t1, n=3)
        assert s0["num_successful_steps"] == s1["num_successful_steps"]


def test_edge_cases(hip, oracle):
    w = synth.make_window(7, num_lines=50)
    # (a) nothing free: costs are the fixed cost, parameters untouched
    w_all = dict(w, fixed_index=np.ones_like(w["fixed_index"]))
    x, s, t = hip.lba_solve(w_all)
    xo, so, _ = oracle.lba_solve(w_all)
    assert np.array_equal(x, w["parameters"]) and s["num_free_parameters"] == 0
    assert abs(s["initial_cost"] - so["initial_cost"]) < 1e-13 * so["initial_cost"] and s["initial_cost"] == s["final_cost"]
    # (b) zero iterations
    x, s, t = hip.lba_solve(w, max_num_iterations=0)
    assert np.array_equal(x, w["parameters"]) and len(t) == 1
    # (c) unused camera / line slots and a ragged, unsorted observation order
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(w["camera_index"]))
    w2 = dict(w, num_cameras=22, num_lines=53,
              camera_index=w["camera_index"][perm], line_index=w["line_index"][perm],
              fixed_index=w["fixed_index"].reshape(-1, 2)[perm].reshape(-1), observations=w["observations"][perm],
              parameters=np.concatenate([w["parameters"][:120], rng.normal(size=12), w["parameters"][120:], rng.uniform(0.2, 1, 12)]))
    x0, s0, t0 = oracle.lba_solve(w2, linear_solver=1)
    x1, s1, t1 = hip.lba_solve(w2)
    _assert_trace_parity(t0, t1, n=3)
    assert np.array_equal(x1[120:132], w2["parameters"][120:132]) and np.array_equal(x1[-12:], w2["parameters"][-12:])
    assert np.abs(x0 - x1).max() < 1e-5
    # (d) a camera observing the same line twice (duplicate residual blocks)
    dup = np.arange(len(w["camera_index"]))
    dup = np.concatenate([dup, dup[w["camera_index"] < 3][:40]])
    w3 = dict(w, camera_index=w["camera_index"][dup], line_index=w["line_index"][dup],
              fixed_index=w["fixed_index"].reshape(-1, 2)[dup].reshape(-1), observations=w["observations"][dup])
    x0, s0, t0 = oracle.lba_solve(w3, linear_solver=1, max_num_iterations=2)
    x1, s1, t1 = hip.lba_solve(w3, max_num_iterations=2)
    _assert_trace_parity(t0, t1)
    # (e) empty window
    we = dict(num_cameras=2, num_lines=2, camera_index=np.zeros(0, np.int32), line_index=np.zeros(0, np.int32),
              fixed_index=np.zeros(0, np.int32), observations=np.zeros((0, 8)), parameters=np.arange(20.0))
    x, s, t = hip.lba_solve(we)
    assert np.array_equal(x, np.arange(20.0)) and s["initial_cost"] == 0.0


def test_golden_optimum(hip):
    """Converged HIP solve vs the scipy.optimize.least_squares optimum committed in tests/golden."""
    z = np.load(os.path.join(GOLD, "lba_optimum.npz"))
    w = {k: z[k] for k in z.files}
    w["num_cameras"], w["num_lines"] = int(z["num_cameras"]), int(z["num_lines"])
    x, s, t = hip.lba_solve(w, max_num_iterations=60, function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-16)
    assert abs(s["final_cost"] - float(z["optimum_cost"])) < 1e-9 * float(z["optimum_cost"])
    assert np.abs(x - z["optimum"]).max() < 1e-4
    assert abs(s["initial_cost"] - float(z["initial_cost"])) < 1e-13 * float(z["initial_cost"])


def test_long_solve_stops_early_like_the_reference_study(hip, oracle):
    """max_num_iter = 1000 (the reference's *_maxnumiter1000 runs): the solve terminates by a
    tolerance long before the cap and the host stops enqueueing iterations."""
    w = synth.make_window(17, num_lines=120)
    x0, s0, t0 = oracle.lba_solve(w, linear_solver=1, max_num_iterations=1000)
    x1, s1, t1 = hip.lba_solve(w, max_num_iterations=1000)
    assert s0["termination_type"] == s1["termination_type"] != 0
    assert abs(s0["num_successful_steps"] - s1["num_successful_steps"]) <= 1
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-6 * s0["final_cost"]
    assert np.abs(x0 - x1).max() < 1e-4


def test_batch_equals_single_and_is_reproducible(hip, oracle):
    """Windows are independent: a window solved inside a ragged batch gives the same result as
    alone; two runs of the same batch are bitwise identical (ordered reductions)."""
    ws = [synth.make_window(20 + i, num_lines=l, num_kf=k, num_free=f)
          for i, (l, k, f) in enumerate([(120, 20, 10), (40, 6, 3), (300, 20, 10), (75, 12, 5), (200, 20, 10), (10, 4, 2)])]
    ws.append(synth.make_motion_only(9, num_lines=30))
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize()
    b.solve(); b.download()
    first = [b.parameters(i).copy() for i in range(len(ws))]
    summ = [b.summary(i) for i in range(len(ws))]
    b.reset(); b.solve(); b.download()
    for i, w in enumerate(ws):
        assert np.array_equal(first[i], b.parameters(i)), "window %d not reproducible" % i
        xs, ss, _ = hip.lba_solve(w)
        assert ss["num_successful_steps"] == summ[i]["num_successful_steps"]
        assert np.abs(xs - first[i]).max() < 1e-9
        xo, so, _ = oracle.lba_solve(w, linear_solver=1)
        assert np.abs(xo - first[i]).max() < 1e-5
        assert abs(so["final_cost"] - summ[i]["final_cost"]) <= 1e-7 * so["final_cost"]
    its = sum(s["num_successful_steps"] + s["num_unsuccessful_steps"] for s in summ)
    assert b.iterations(clear=True) == 2 * its          # two solves since finalize
    assert b.counts()["windows"] == len(ws)
    b.close()


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_matrix_core_elimination_sweep(hip, oracle, mode):
    """lba_elimination = 2 / 3 / 4 (4: group-local accumulators, lba_eliminate_grouped.h, lines packed by first free camera): Schur outer products on v_mfma_f64_16x16x4_f64 with the accumulator tiles in registers
    (lba_eliminate_mfma.h), normal-equation blocks from the Gram formulation (lba_gram.h), camera constants applied by the
    reduced solve.  Different operation order than the default sweep and the oracle, the same algebra: the first three
    iterations agree with the oracle to the trace tolerances above, step counts and terminations are equal, final cost
    to 1e-6 relative and parameters to 1e-5 (the late iterations inherit the conditioning of the problem, see the header
    of this file); two runs are bitwise identical.  Shapes: the bench window family, few cameras, tiles with more than
    32 lines (sources derived from the line masks), lines that span several lane rows, motion-only (no free line)."""
    shapes = [dict(num_lines=150), dict(num_lines=500), dict(num_lines=60, num_kf=6, num_free=3),
              dict(num_lines=90, num_kf=8, num_free=6, mean_track=2.0),
              dict(num_lines=60, num_kf=24, num_free=10, mean_track=40.0)]
    ws = [synth.make_window(70 + i, **kw) for i, kw in enumerate(shapes)]
    ws.append(synth.make_motion_only(5, num_lines=40))
    for w in ws:
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = hip.lba_solve(w, lba_elimination=mode, lba_fused_motion_only=0)
        xd, sd, td = hip.lba_solve(w, lba_elimination=1, lba_fused_motion_only=0)
        _assert_trace_parity(t0, t1, n=3)
        for k in ("num_successful_steps", "num_unsuccessful_steps", "termination_type", "num_free_parameters", "num_residual_blocks"):
            assert s0[k] == s1[k] == sd[k], k
        assert abs(s0["initial_cost"] - s1["initial_cost"]) <= 1e-12 * s0["initial_cost"]
        assert abs(s0["final_cost"] - s1["final_cost"]) <= 1e-6 * s0["final_cost"]
        assert np.abs(x0 - x1).max() < 1e-5 and np.abs(xd - x1).max() < 1e-5
    b = hip.LBABatch()
    for w in ws[:4]:
        b.add(w)
    b.finalize(lba_elimination=mode)
    b.solve(); b.download()
    first = [b.parameters(i).copy() for i in range(4)]
    b.reset(); b.solve(); b.download()
    for i in range(4):
        assert np.array_equal(first[i], b.parameters(i)), "window %d not reproducible" % i
        xs, _, _ = hip.lba_solve(ws[i], lba_elimination=mode)
        assert np.abs(xs - first[i]).max() < 1e-8          # batch vs single: different chunking, same algebra
    b.close()


def test_spilled_elimination_variant(hip, oracle):
    """reuse_elimination = 1: the back-substitution streams the F blocks the elimination kernel
    spilled instead of re-linearising; same algebra, so the same solve to round-off."""
    w = synth.make_window(31, num_lines=250)
    xa, sa, ta = hip.lba_solve(w)
    xb, sb, tb = hip.lba_solve(w, reuse_elimination=1)
    _assert_trace_parity(ta, tb, n=4)
    assert sa["num_successful_steps"] == sb["num_successful_steps"]
    assert np.abs(xa - xb).max() < 1e-7
    xo, so, _ = oracle.lba_solve(w, linear_solver=1)
    assert np.abs(xb - xo).max() < 1e-5


def test_graph_replay_equals_eager_launches(hip):
    w = [synth.make_window(40 + i, num_lines=100) for i in range(4)]
    out = []
    for use_graph in (1, 0):
        b = hip.LBABatch()
        for x in w:
            b.add(x)
        b.finalize(use_graph=use_graph)
        b.solve()
        b.download()
        out.append([b.parameters(i) for i in range(4)])
        if not use_graph:
            b.set_profiling(True)
            b.reset(); b.solve(); b.download()
            kt = b.kernel_times()
            assert kt["linearise_schur"][1] == 10 and kt["linearise_schur"][0] > 0
            assert all(np.array_equal(b.parameters(i), out[-1][i]) for i in range(4))
        b.close()
    for a, c in zip(*out):
        assert np.array_equal(a, c)


def test_full_size_window_properties(hip, oracle):
    """BASELINE config: 10 free + 10 fixed keyframes, 2000 lines.  Oracle parity on the full window
    plus size-independent properties: monotone accepted costs, fixed blocks untouched, the optimum
    reduces the reprojection cost to the noise floor, trajectory error vs truth shrinks."""
    w = synth.make_window(1234, num_lines=2000)
    x1, s1, t1 = hip.lba_solve(w)
    x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
    _assert_trace_parity(t0, t1, n=3, tight=True)      # all 10 iterations: same accept / reject decisions, measured tolerances
    _assert_params_parity(w, x0, x1)
    assert s0["num_successful_steps"] == s1["num_successful_steps"]
    assert abs(s0["final_cost"] - s1["final_cost"]) <= 3e-7 * s0["final_cost"]
    costs = [r["cost"] for r in t1]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
    assert np.array_equal(x1[60:120], w["parameters"][60:120])           # the 10 fixed keyframes
    M = len(w["camera_index"])
    noise_floor = 0.5 * 4 * M * (0.5 / 406.05) ** 2
    assert s1["final_cost"] < 1.5 * noise_floor < s1["initial_cost"]
    c_true = synth.camera_centers(w["true_parameters"][:60])
    e0 = np.linalg.norm(synth.camera_centers(w["parameters"][:60]) - c_true, axis=1)
    e1 = np.linalg.norm(synth.camera_centers(x1[:60]) - c_true, axis=1)
    assert np.sqrt((e1 ** 2).mean()) < np.sqrt((e0 ** 2).mean())
    eo = np.linalg.norm(synth.camera_centers(x1[:60]) - synth.camera_centers(x0[:60]), axis=1)
    assert np.sqrt((eo ** 2).mean()) < 1e-9                              # trajectory RMS vs the oracle solve (measured 1e-10 m)


def test_headline_path_matches_oracle(hip, oracle):
    """The code path bench.py's `value` is measured on, under the oracle: the bench's own batch - 1024 windows
    of 2000 lines (synth.make_window(0 .. 1023), BASELINE config 3), DEFAULT options - so that the automatic choices are the headline's:
    the grouped matrix-core sweep (slslam_lba_batch_elimination = 4) on graded chunk cuts (slslam_lba_batch_window_chunks < 0), hipGraph
    replay.  32 windows spread over the batch against oracle.lba_solve(linear_solver = 1): the full iteration trace at the file's TIGHT
    tolerances (same accept / reject decisions at every iteration), summary, camera AND line parameters; plus: the batch is reproducible
    bit for bit, and a window of it equals the window alone with the reported cut and sweep.  What LBAProblem::build + ceres::Solve do
    per window at reference src/slam.cpp:924-952."""
    B = int(os.environ.get("SLSLAM_HEADLINE_WINDOWS", "1024"))
    ws = [synth.make_window(i, num_lines=2000) for i in range(B)]
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize()                                       # every option at its default, as bench.py runs it
    assert b.elimination() == 4, "the headline batch is expected to take the grouped matrix-core sweep"
    assert b.path() == 0
    picks = sorted(set(int(round(x)) for x in np.linspace(0, B - 1, 32)))
    cuts = [b.window_chunks(i) for i in picks]
    assert all(c < 0 for c in cuts), cuts              # graded chunk sizes: -(1000 rounds + chunks)
    b.solve(); b.download()
    got = {i: (b.parameters(i).copy(), b.summary(i), b.trace(i)) for i in picks}
    its = b.iterations(clear=True)
    b.reset(); b.solve(); b.download()
    assert b.iterations(clear=True) == its
    for i in picks:
        assert np.array_equal(got[i][0], b.parameters(i)), "window %d of the headline batch is not reproducible" % i
    b.close()
    # Tolerances.  TIGHT (this file's header) was measured on twelve windows; over the bench's own population a few windows are
    # ill-conditioned enough (depth-degenerate lines: an eigenvalue of their 4 x 4 block ~1e-8 of the others) that ANY change of
    # summation order moves their late iterations beyond it - the oracle itself does when its initial parameters are perturbed by
    # 1e-15 / 1e-13 relative (window 99: step norm 1.3e-6 / 2.1e-4, line parameters 6.6e-6 / 1.1e-3; profiles/round5_headline_parity_study.txt
    # has all 32 windows for this sweep and for the LDS-atomic one, which exceeds TIGHT on the same windows).  So: identical accept /
    # reject decisions and summaries on EVERY window; TIGHT on at least 80 % of them; every window within CAP = ~10 x the worst deviation
    # measured over the 32 windows for either sweep (cost 9e-8, radius 3.9e-6, step norm 2.3e-5, gain ratio 1.1e-4, cameras 2.9e-10,
    # lines 1.2e-4); for a window beyond TIGHT the oracle's own movement under the two perturbations is printed beside it.
    CAP = dict(cost=1e-6, radius=5e-5, step_norm=3e-4, rho=1e-3, cam=3e-9, line=1.5e-3)

    def deviations(w, xa, ta, xb, tb):
        nc = 6 * int(w["num_cameras"])
        d = dict(cost=0.0, radius=0.0, step_norm=0.0, rho=0.0)
        for a, c in zip(ta, tb):
            d["cost"] = max(d["cost"], abs(a["cost"] - c["cost"]) / abs(a["cost"]))
            d["radius"] = max(d["radius"], abs(a["trust_region_radius"] - c["trust_region_radius"]) / a["trust_region_radius"])
            d["step_norm"] = max(d["step_norm"], abs(a["step_norm"] - c["step_norm"]) / (a["step_norm"] + 1e-12))
            d["rho"] = max(d["rho"], abs(a["relative_decrease"] - c["relative_decrease"]) / (abs(a["relative_decrease"]) + 1e-3))
        d["cam"] = float(np.abs(xa[:nc] - xb[:nc]).max())
        d["line"] = float(np.abs(xa[nc:] - xb[nc:]).max())
        return d

    worst = {k: 0.0 for k in CAP}
    rejected, beyond_tight = 0, []
    for i in picks:
        w = ws[i]
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        x1, s1, t1 = got[i]
        assert len(t0) == len(t1)
        for a, c in zip(t0, t1):                       # the same path through the trust-region policy, iteration by iteration
            assert a["iteration"] == c["iteration"] and a["step_is_successful"] == c["step_is_successful"] and a["step_is_valid"] == c["step_is_valid"]
        _assert_trace_parity(t0, t1, n=2)              # initial evaluation and first step: round-off
        _assert_summary_parity(s0, s1)
        rejected += s1["num_unsuccessful_steps"]
        d = deviations(w, x0, t0, x1, t1)
        for k in worst:
            worst[k] = max(worst[k], d[k])
            assert d[k] <= CAP[k], (i, k, d[k])
        if any(d[k] > TIGHT[k] for k in TIGHT):
            yard = {k: 0.0 for k in CAP}
            for eps in (1e-15, 1e-13):
                rng = np.random.default_rng(i)
                pert = np.array(w["parameters"], dtype=np.float64) * (1.0 + eps * rng.choice([-1.0, 1.0], size=len(w["parameters"])))
                xp, sp, tp = oracle.lba_solve(dict(w, parameters=pert), linear_solver=1)
                if len(tp) == len(t0):
                    dp = deviations(w, x0, t0, xp, tp)
                    yard = {k: max(yard[k], dp[k]) for k in yard}
                else:
                    yard = {k: float("inf") for k in yard}      # the perturbed oracle run even takes another number of iterations
            # (printed, not asserted: how far one random perturbation moves the oracle is a noisy yardstick - window 660 moves the HIP path
            # 4e-6 in step norm and the oracle 2e-7 under THIS sign pattern, 2.3e-6 under the LDS-atomic sweep's summation order)
            beyond_tight.append((i, {k: "%.1e (oracle under perturbation %.1e)" % (d[k], yard[k]) for k in d if d[k] > TIGHT[k]}))
    print("headline path vs oracle over %d windows of %d: worst deviations %s; beyond TIGHT: %s" % (len(picks), B, worst, beyond_tight))
    assert len(beyond_tight) <= len(picks) // 5
    assert rejected > 0                                # the bench family rejects about a third of its steps: both branches of the policy ran
    for i, cut in list(zip(picks, cuts))[:4]:          # a window's bytes are a function of the window, the sweep and the cut
        xs, ss, ts = hip.lba_solve(ws[i], lba_elimination=4, chunks_per_window=cut)
        assert np.array_equal(xs, got[i][0]) and ss == got[i][1]


def test_mixed_precision_solves(hip, oracle):
    """slslam_solver_options.lba_precision = 1 (VERDICT round 4, row *): the steady elimination sweeps form the CAMERA Jacobian of an
    observation in float (packed fp32 on gfx950), park it in LDS as floats and widen it where it is used; geometry, residuals, the line
    Jacobian, every block product and sum, the candidate evaluation and the trust-region bookkeeping are the double path's (reference
    arithmetic: src/lba_problem.h:46-118 on Jet<double>).  Which parts can be float was measured (tools/mixed_precision_study.py,
    profiles/round5_mixed_precision_study.txt): with the line Jacobian in float as well, 8 of 28 windows change an accept / reject decision.
    STATED TOLERANCE against the double path and against the oracle, bench family (2000 / 500 / 150 lines) and a window of long
    tracks: the same accept / reject sequence, final cost 3e-4 relative (measured 1.2e-4), every iteration's cost 1e-3 relative (1.8e-4), camera poses 1e-4 (8e-7)
    (rad / m); line closest points within 10 m: 1e-3 m for at least 99 % of a window's lines (the others are depth-degenerate lines -
    seen under a few degrees of parallax, a flat direction of the cost - and stay within 1 m).  Opt-in, never the bench's `value`;
    asking for it where the grouped sweep cannot run is refused."""
    ws = [synth.make_window(i, num_lines=n) for i, n in ((0, 2000), (1, 2000), (1234, 2000), (5, 500), (7, 500), (11, 150), (12, 150), (205, 150))]
    ws.append(synth.make_window(75, num_lines=60, num_kf=24, num_free=10, mean_track=40.0))
    worst = dict(final_cost=0.0, cam=0.0, cp_p99=0.0, cp_max=0.0, iter_cost=0.0)
    for w in ws:
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        xd, sd, td = hip.lba_solve(w, lba_elimination=4)
        xm, sm, tm = hip.lba_solve(w, lba_precision=1)
        nc = 6 * int(w["num_cameras"])
        for ref_x, ref_s, ref_t in ((xd, sd, td), (x0, s0, t0)):
            assert sm["num_successful_steps"] == ref_s["num_successful_steps"] and sm["num_unsuccessful_steps"] == ref_s["num_unsuccessful_steps"]
            assert sm["termination_type"] == ref_s["termination_type"]
            assert abs(sm["initial_cost"] - ref_s["initial_cost"]) <= 1e-12 * ref_s["initial_cost"]      # the first sweep is double
            worst["final_cost"] = max(worst["final_cost"], abs(sm["final_cost"] - ref_s["final_cost"]) / ref_s["final_cost"])
            worst["cam"] = max(worst["cam"], float(np.abs(xm[:nc] - ref_x[:nc]).max()))
            cpm = np.array([synth.orth_to_av(u)[:3] for u in xm[nc:].reshape(-1, 4)])
            cpr = np.array([synth.orth_to_av(u)[:3] for u in ref_x[nc:].reshape(-1, 4)])
            near = np.linalg.norm(cpr, axis=1) < 10.0                # (closest points at <= 10 m depth)
            dcp = np.abs(cpm - cpr)[near].max(axis=1)
            worst["cp_p99"] = max(worst["cp_p99"], float(np.percentile(dcp, 99)))
            worst["cp_max"] = max(worst["cp_max"], float(dcp.max()))
            assert len(tm) == len(ref_t)
            for a, c in zip(ref_t, tm):
                assert a["step_is_successful"] == c["step_is_successful"] and a["step_is_valid"] == c["step_is_valid"]
                worst["iter_cost"] = max(worst["iter_cost"], abs(a["cost"] - c["cost"]) / abs(a["cost"]))
    print("mixed precision vs double / oracle over %d windows: %s" % (len(ws), worst))
    assert worst["final_cost"] <= 3e-4 and worst["cam"] <= 1e-4 and worst["iter_cost"] <= 1e-3
    assert worst["cp_p99"] <= 1e-3 and worst["cp_max"] <= 1.0
    # a batch in mixed precision: reproducible bit for bit, equal to its windows alone
    b = hip.LBABatch()
    for w in ws[3:7]:
        b.add(w)
    b.finalize(lba_precision=1)
    assert b.elimination() == 4
    b.solve(); b.download()
    first = [b.parameters(i).copy() for i in range(4)]
    b.reset(); b.solve(); b.download()
    for i in range(4):
        assert np.array_equal(first[i], b.parameters(i))
        xs, _, _ = hip.lba_solve(ws[3 + i], lba_precision=1)
        assert np.abs(xs - first[i]).max() < 1e-6
    b.close()
    wide = synth.make_window(41, num_lines=60, num_kf=30, num_free=14)
    for bad in (dict(lba_precision=1, lba_elimination=1), dict(lba_precision=2)):
        with pytest.raises(hip.SlslamError):
            hip.lba_solve(ws[5], **bad)
    with pytest.raises(hip.SlslamError) as e:
        hip.lba_solve(wide, lba_precision=1)               # 14 free cameras: the grouped sweep cannot take it
    assert e.value.status == 4


def test_batched_motion_only(hip, oracle):
    """motion_only_ba (reference src/slam.cpp:578-675) batched over frames.  Same
    kernels, degenerate shape: one free camera, every line constant, 6x6 reduced system."""
    ws = [synth.make_motion_only(500 + i, num_lines=60 + (i % 7) * 10) for i in range(96)]
    b = hip.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize()
    b.solve(); b.download()
    for i in (0, 1, 17, 50, 95):
        xo, so, to = oracle.lba_solve(ws[i])
        x, s = b.parameters(i), b.summary(i)
        assert s["num_free_parameters"] == 6 and s["fixed_cost"] > 0
        assert s["num_successful_steps"] == so["num_successful_steps"]
        assert abs(s["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
        assert np.abs(x[:6] - xo[:6]).max() < 1e-9 and np.array_equal(x[6:], ws[i]["parameters"][6:])
    b.close()


def test_randomised_shapes_against_oracle(hip, oracle):
    """60 random small windows (2..24 keyframes, 2..20 free, 3..120 lines, track lengths, noise levels, scrambled
    observation order, constant lines, loss on / off, iteration caps): initial evaluation, program reduction and
    the first iterations agree with the oracle on every one (tests/tools/fuzz_parity.py, run with 800 cases in round 1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(60, seed=3, verbose=True)
    assert bad == 0
    assert worst["trace"] < 1e-6
    # ... and 18 random windows beyond the tiled sweeps (65-90 keyframes, 21-45 free cameras, lines tracked through more than
    # 64 keyframes; scrambled order, constant lines, loss on / off, iteration caps): the global-memory path
    bad, worst = mod.run(18, seed=5, verbose=True, oversize=True)
    assert bad == 0
    assert worst["trace"] < 1e-6


def test_randomised_shapes_grouped_matrix_core_sweep(hip, oracle):
    """The same 60 random windows through lba_elimination = 4 (lines packed by first free camera, group-local accumulator tiles,
    lba_eliminate_grouped.h; windows with more than 10 free cameras or a camera that sees a line twice take the default sweep):
    scrambled observation order, track lengths of 2 (camera ranges with holes), constant lines, motion-only shapes,
    loss on / off, iteration caps."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(60, seed=3, verbose=True, hip_opt=dict(lba_elimination=4, lba_fused_motion_only=0))
    assert bad == 0
    assert worst["trace"] < 1e-6


def test_kept_jacobian_sweep_after_rejected_steps(hip, oracle):
    """lba_keep_jacobian (grouped sweep): after a REJECTED step the next elimination sweep does not linearise - it replays the
    blocks J_c'^T J_l and the line blocks the last linearising sweep left in memory, with the new radius (what
    ceres::TrustRegionMinimizer does: the Jacobian is evaluated after successful steps only).  Windows whose solves reject steps
    (the bench family rejects about a third, several in a row among them): the iteration traces with and without the replay
    agree to round-off at every iteration, decisions and terminations are equal, both match the oracle, a batch is reproducible
    bit for bit and equal to its windows solved alone."""
    ws = [synth.make_window(i, num_lines=n) for i, n in ((0, 400), (1, 400), (3, 250), (7, 600), (11, 150))]
    ws.append(synth.make_window(75, num_lines=60, num_kf=24, num_free=10, mean_track=40.0))
    rejected = 0
    for w in ws:
        x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
        xk, sk, tk = hip.lba_solve(w, lba_elimination=4, lba_keep_jacobian=1)
        xn, sn, tn = hip.lba_solve(w, lba_elimination=4, lba_keep_jacobian=0)
        rejected += sk["num_unsuccessful_steps"]
        assert len(tk) == len(tn) == len(t0)
        for a, b, c in zip(tk, tn, t0):
            assert a["step_is_successful"] == b["step_is_successful"] == c["step_is_successful"]
            assert abs(a["cost"] - b["cost"]) <= 1e-10 * abs(b["cost"]) + 1e-300
            assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-7 * b["trust_region_radius"]
        _assert_trace_parity(t0, tk, n=3)
        for k in ("num_successful_steps", "num_unsuccessful_steps", "termination_type"):
            assert s0[k] == sk[k] == sn[k], k
        assert abs(sk["final_cost"] - sn["final_cost"]) <= 1e-9 * sn["final_cost"]
        assert np.abs(xk - xn).max() < 1e-7 and np.abs(xk - x0).max() < 1e-5
    assert rejected >= 6, "the shapes of this test are meant to reject steps"
    b = hip.LBABatch()
    for w in ws[:5]:
        b.add(w)
    b.finalize(lba_elimination=4, lba_keep_jacobian=1)
    assert b.elimination() == 4
    b.solve(); b.download()
    first = [b.parameters(i).copy() for i in range(5)]
    b.reset(); b.solve(); b.download()
    for i in range(5):
        assert np.array_equal(first[i], b.parameters(i)), "window %d not reproducible" % i
        xs, _, _ = hip.lba_solve(ws[i], lba_elimination=4, lba_keep_jacobian=1)
        assert np.abs(xs - first[i]).max() < 1e-8
    b.close()


def test_graded_chunk_sizes(hip, oracle):
    """chunks_per_window = -(1000 r + c): c chunks of graded sizes made for r rounds of the wave slots (what the automatic choice cuts the
    windows of a chip-filling batch with, reported by slslam_lba_batch_window_chunks; long chunks first in the dispatch order).  Forced on
    small windows: same LM decisions and results as the equal cut to round-off, parity with the oracle, bitwise reproducible, a window
    in a batch equals the window alone with the same cut, both sweeps; malformed requests are refused."""
    ws = [synth.make_window(60 + i, num_lines=n) for i, n in enumerate((500, 800, 650))]
    for elim in (1, 4):
        for cut in (-2004, -3006, -3007):
            b = hip.LBABatch()
            for w in ws:
                b.add(w)
            b.finalize(lba_elimination=elim, chunks_per_window=cut)
            assert [b.window_chunks(i) for i in range(3)] == [cut] * 3
            b.solve(); b.download()
            first = [b.parameters(i).copy() for i in range(3)]
            b.reset(); b.solve(); b.download()
            for i, w in enumerate(ws):
                assert np.array_equal(first[i], b.parameters(i))
                xs, ss, ts = hip.lba_solve(w, lba_elimination=elim, chunks_per_window=cut)
                assert np.array_equal(xs, first[i]) and ss == b.summary(i)
                xe, se, te = hip.lba_solve(w, lba_elimination=elim, chunks_per_window=-cut % 1000)
                for k in ("num_successful_steps", "num_unsuccessful_steps", "termination_type"):
                    assert ss[k] == se[k]
                assert abs(ss["final_cost"] - se["final_cost"]) <= 1e-9 * se["final_cost"] and np.abs(xs - xe).max() < 1e-7
            b.close()
    x0, s0, t0 = oracle.lba_solve(ws[0], linear_solver=1)
    x1, s1, t1 = hip.lba_solve(ws[0], chunks_per_window=-3006)
    _assert_trace_parity(t0, t1, n=3); _assert_summary_parity(s0, s1)
    for bad in (-6, -1003, -2001):
        with pytest.raises(hip.SlslamError):
            hip.lba_solve(ws[0], chunks_per_window=bad)


def test_non_finite_input_is_refused_at_the_boundary(hip):
    """ADVICE round 3 (lane_F / line_block rely on K = 0 for constant lines: 0 x inf would be NaN): non-finite observations or
    parameters never reach a kernel - slslam_pack_window / slslam_lba_solve / slslam_lba_batch_add return
    SLSLAM_ERR_INVALID_ARGUMENT (the reference has no error convention, SURVEY 8b: a replacement may add a status) on every
    path (fused motion-only, general sweeps, constant lines among free ones), whichever block holds the value."""
    mo = synth.make_motion_only(12, num_lines=60)
    mixed = dict(synth.make_window(13, num_lines=120))
    fx = np.asarray(mixed["fixed_index"]).reshape(-1, 2).copy()
    fx[np.isin(np.asarray(mixed["line_index"]), (3, 17, 40)), 1] = 1
    mixed["fixed_index"] = fx.reshape(-1)
    for base, opts in ((mo, dict()), (mo, dict(lba_fused_motion_only=0)), (mixed, dict()), (mixed, dict(lba_elimination=4))):
        hip.lba_solve(base, **opts)                            # the finite window solves
        fixed = np.asarray(base["fixed_index"]).reshape(-1, 2)
        for cam_fixed in (0, 1):
            idx = np.flatnonzero((fixed[:, 1] == 1) & (fixed[:, 0] == cam_fixed))
            if len(idx) == 0:
                idx = np.flatnonzero(fixed[:, 0] == cam_fixed)
            i = int(idx[len(idx) // 2])
            for val in (np.inf, -np.inf, np.nan):
                w = dict(base); ob = np.array(w["observations"], dtype=np.float64).reshape(-1, 8).copy()
                ob[i, 3] = val; w["observations"] = ob.reshape(-1)
                with pytest.raises(hip.SlslamError) as e:
                    hip.lba_solve(w, **opts)
                assert e.value.status == 1
                b = hip.LBABatch()
                with pytest.raises(hip.SlslamError):
                    b.add(w)
                b.close()
        w = dict(base); x = np.array(w["parameters"], dtype=np.float64).reshape(-1).copy(); x[7] = np.nan; w["parameters"] = x
        with pytest.raises(hip.SlslamError):
            hip.lba_solve(w, **opts)


def test_one_shot_solves_reuse_their_device_block(hip, oracle):
    """slslam_lba_solve / slslam_po_solve keep the device block of the previous call (device_cache.h): solves of different
    shapes back to back, interleaved with pose-graph solves, still match the oracle - nothing depends on fresh memory."""
    shapes = [dict(num_lines=300), dict(num_lines=40, num_kf=6, num_free=3), dict(num_lines=500), dict(num_lines=60, num_kf=24, num_free=12, mean_track=30.0)]
    g = synth.make_pose_graph(3, num_poses=60, num_loops=4)
    xg0, sg0, _ = oracle.po_solve(g)
    for rep in range(2):
        for i, kw in enumerate(shapes):
            w = synth.make_window(40 + i, **kw)
            x0, s0, t0 = oracle.lba_solve(w, linear_solver=1)
            x1, s1, t1 = hip.lba_solve(w)
            _assert_trace_parity(t0, t1, n=3)
            _assert_summary_parity(s0, s1)
            xg, sg, _ = hip.po_solve(g)
            assert abs(sg["final_cost"] - sg0["final_cost"]) <= 1e-6 * sg0["final_cost"] + 1e-15 and np.abs(xg - xg0).max() < 1e-5
        hip.release_cached_memory()


def test_fused_motion_only_matches_general_path_and_oracle(hip, oracle):
    """Motion-only windows (one free camera, constant lines) are solved by one launch with the 6 x 6 system in registers
    (lba_motion_only.h); same trust-region policy as the general four-launch path: identical step counts and termination,
    poses and traces to round-off - near and far from the optimum, with rejected steps (forced through
    min_relative_decrease), with and without the robust loss, for 0, 1 and many iterations."""
    cases = [(0, {}), (1, {"max_num_iterations": 0}), (2, {"max_num_iterations": 1}), (3, {"max_num_iterations": 40}),
             (4, {"huber_delta": 0.0}), (5, {"huber_delta": 0.0, "min_relative_decrease": 1.5}),
             (6, {"huber_delta": 0.0, "min_relative_decrease": 1.0000005}), (7, {"jacobi_scaling": 0}), (8, {})]
    rejected = 0
    for seed, kw in cases:
        w = synth.make_motion_only(800 + seed, num_lines=30 + 25 * seed, noise_px=[0.0, 0.5, 2.0][seed % 3])
        if seed % 2:
            rng = np.random.default_rng(seed)
            w["parameters"] = w["parameters"].copy()
            w["parameters"][:6] += rng.normal(0, 1.0, 6) * [0.05, 0.05, 0.05, 0.5, 0.5, 0.5]
        x0, s0, t0 = hip.lba_solve(w, lba_fused_motion_only=0, **kw)
        x1, s1, t1 = hip.lba_solve(w, lba_fused_motion_only=1, **kw)
        okw = {k: v for k, v in kw.items() if k != "huber_delta"}
        xo, so, to = oracle.lba_solve(w, huber_delta=kw.get("huber_delta", 1.0 / 406.05), **okw)
        for s in (s0, so):
            _assert_summary_parity(s, s1)
        _assert_trace_parity(t0, t1)
        _assert_trace_parity(to, t1)
        assert np.abs(x0 - x1).max() < 1e-9 and np.abs(xo - x1).max() < 1e-7
        rejected += s1["num_unsuccessful_steps"]
    assert rejected > 0
    # batches that mix a motion-only window with an ordinary one take the general path and still solve both
    b = hip.LBABatch()
    wm, wg = synth.make_motion_only(900, num_lines=40), synth.make_window(901, num_lines=60)
    b.add(wm); b.add(wg)
    b.finalize()
    b.solve(); b.download()
    for i, w in enumerate((wm, wg)):
        xo, so, _ = oracle.lba_solve(w, linear_solver=1)
        assert np.abs(b.parameters(i) - xo).max() < 1e-5
    b.close()
