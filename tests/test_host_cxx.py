"""The host-side C++ mirror of the reference classes (slslam_amd/host): compiles the reference's
call protocol against it (CPU), and runs it end to end on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from slslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slslam_amd", "host")
LIBDIR = os.path.join(ROOT, "slslam_amd", "_lib")
DEMO = os.path.join(ROOT, "tests", "_build", "drop_in_demo")


def _build_demo():
    subprocess.check_call(["make", "-s", "-C", HOST])
    os.makedirs(os.path.dirname(DEMO), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", HOST, "-o", DEMO,
                           os.path.join(ROOT, "tests", "host_cxx", "drop_in_demo.cpp"),
                           "-L", LIBDIR, "-lslslam_host", "-lslslam_hip", "-Wl,-rpath," + LIBDIR])
    return DEMO


def _write_lba(path, w, iters=10, robust=1):
    with open(path, "wb") as f:
        np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"]), iters, robust], dtype=np.int32).tofile(f)
        np.asarray(w["camera_index"], dtype=np.int32).tofile(f)
        np.asarray(w["line_index"], dtype=np.int32).tofile(f)
        np.asarray(w["fixed_index"], dtype=np.int32).tofile(f)
        np.asarray(w["observations"], dtype=np.float64).tofile(f)
        np.asarray(w["parameters"], dtype=np.float64).tofile(f)


def test_reference_call_protocol_compiles_and_fails_loudly_without_gpu(tmp_path):
    """The call sites of slam.cpp compile against the mirrored headers; on a box without a GPU the
    solve reports the back-end error instead of silently doing nothing (no CPU fallback)."""
    demo = _build_demo()
    from slslam_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    w = synth.make_window(3, num_lines=20)
    _write_lba(tmp_path / "in.bin", w)
    p = subprocess.run([demo, "lba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert p.returncode == 2                      # SLSLAM_ERR_NO_DEVICE
    assert "no usable HIP device" in p.stderr
    out = np.fromfile(tmp_path / "out.bin")
    assert np.array_equal(out[:-5], w["parameters"])          # parameters untouched


@pytest.mark.gpu
def test_drop_in_lba_and_po_on_gpu(tmp_path, hip, oracle):
    demo = _build_demo()
    w = synth.make_window(5, num_lines=150)
    _write_lba(tmp_path / "in.bin", w)
    subprocess.check_call([demo, "lba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    out = np.fromfile(tmp_path / "out.bin")
    x, tail = out[:-5], out[-5:]
    xs, ss, _ = hip.lba_solve(w)
    assert np.array_equal(x, xs)                              # same C ABI underneath
    assert (tail[0], tail[1]) == (ss["num_successful_steps"], ss["num_unsuccessful_steps"]) and tail[4] == 0
    xo, so, _ = oracle.lba_solve(w, linear_solver=1)
    assert np.abs(x - xo).max() < 1e-5 and abs(tail[3] - so["final_cost"]) < 1e-7 * so["final_cost"]
    # motion_only_ba shape through the same classes, non-robust flag honoured
    m = synth.make_motion_only(6, num_lines=40)
    _write_lba(tmp_path / "in2.bin", m, robust=0)
    subprocess.check_call([demo, "lba", str(tmp_path / "in2.bin"), str(tmp_path / "out2.bin")])
    x2 = np.fromfile(tmp_path / "out2.bin")[:-5]
    xo2, _, _ = oracle.lba_solve(m, huber_delta=0.0)
    assert np.abs(x2 - xo2).max() < 1e-8 and np.array_equal(x2[6:], m["parameters"][6:])
    # pose graph
    g = synth.make_pose_graph(2, num_poses=50, num_loops=3)
    with open(tmp_path / "po.bin", "wb") as f:
        np.array([g["num_poses"], len(g["pose_index_1"])], dtype=np.int32).tofile(f)
        g["pose_index_1"].astype(np.int32).tofile(f); g["pose_index_2"].astype(np.int32).tofile(f)
        np.asarray(g["constraints"], dtype=np.float64).tofile(f); np.asarray(g["parameters"], dtype=np.float64).tofile(f)
    subprocess.check_call([demo, "po", str(tmp_path / "po.bin"), str(tmp_path / "po_out.bin")])
    xp = np.fromfile(tmp_path / "po_out.bin")[:-5]
    xq, sq, _ = oracle.po_solve(g)
    assert np.abs(xp - xq).max() < 1e-6


def test_gc_boundary_encodings_match_numpy_restatement():
    """gc_lite (C++ host library) against the numpy restatements in slslam_amd/synth.py and the
    oracle's copy: pose <-> (w,t), SE(3) algebra, line transforms, orthonormal encode/decode."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", HOST])
    lib = C.CDLL(os.path.join(LIBDIR, "libslslam_host.so"))

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]
    dp = C.POINTER(C.c_double)

    def p(a):
        return a.ctypes.data_as(dp)
    rng = np.random.default_rng(5)
    for scale in (0.0, 1e-9, 0.3, 2.5, 3.1):
        wt = np.concatenate([rng.normal(size=3), rng.normal(size=3)])
        wt[:3] *= scale / max(np.linalg.norm(wt[:3]), 1e-300) if scale > 0 else 0.0
        T = Pose()
        lib.slslam_gc_wt_to_Rt(p(wt), C.byref(T))
        R = np.array(T.R).reshape(3, 3)
        assert np.abs(R - synth.rodrigues(wt[:3])).max() < 1e-14
        back = np.zeros(6)
        lib.slslam_gc_Rt_to_wt(C.byref(T), p(back))
        assert np.abs(back - wt).max() < 1e-9
        Ti, I2 = Pose(), Pose()
        lib.slslam_gc_T_inv(C.byref(T), C.byref(Ti))
        lib.slslam_gc_T_20(C.byref(T), C.byref(Ti), C.byref(I2))
        assert np.abs(np.array(I2.R).reshape(3, 3) - np.eye(3)).max() < 1e-14 and np.abs(np.array(I2.t)).max() < 1e-14
        T2 = Pose()
        wt2 = rng.normal(size=6)
        lib.slslam_gc_wt_to_Rt(p(wt2), C.byref(T2))
        T21 = Pose()
        lib.slslam_gc_T_21(C.byref(T2), C.byref(T), C.byref(T21))       # T21 = T2 * T^-1
        chk = Pose()
        lib.slslam_gc_T_20(C.byref(T21), C.byref(T), C.byref(chk))
        assert np.abs(np.array(chk.R) - np.array(T2.R)).max() < 1e-13 and np.abs(np.array(chk.t) - np.array(T2.t)).max() < 1e-13
        # lines
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        pt = rng.normal(size=3) * 3
        av = np.concatenate([pt - (pt @ d) * d, d])
        orth, av2 = np.zeros(4), np.zeros(6)
        lib.slslam_gc_av_to_orth(p(av), p(orth))
        lib.slslam_gc_orth_to_av(p(orth), p(av2))
        assert np.abs(orth - synth.av_to_orth(av)).max() < 1e-14 and np.abs(av2 - av).max() < 1e-12
        lc, lw = np.zeros(6), np.zeros(6)
        lib.slslam_gc_line_to_pose(p(av), C.byref(T), p(lc))
        assert np.abs(lc[:3] - (R @ av[:3] + wt[3:])).max() < 1e-14 and np.abs(lc[3:] - R @ av[3:]).max() < 1e-14
        lib.slslam_gc_line_from_pose(p(lc), C.byref(T), p(lw))
        assert np.abs(lw - av).max() < 1e-13
