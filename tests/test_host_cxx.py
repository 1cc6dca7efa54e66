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
