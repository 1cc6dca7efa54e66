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
# (tests/test_sanitizers.py points this at the -fsanitize=address,undefined build of the same sources)
HOST_LIB = os.environ.get("SLSLAM_HOST_LIB") or os.path.join(LIBDIR, "libslslam_host.so")
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
    lib = C.CDLL(HOST_LIB)

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


def _map_from_window(w, lib, C):
    """A keyframe / landmark map (the reference's kfs / lms / ba_kfs) that should pack to window w."""
    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    class KF(C.Structure):
        _fields_ = [("id", C.c_int), ("ba_rank", C.c_int), ("T", Pose), ("member_lms", C.POINTER(C.c_int)), ("num_member_lms", C.c_int)]

    class Obs(C.Structure):
        _fields_ = [("kf_id", C.c_int), ("obs", C.c_double * 8)]

    class LM(C.Structure):
        _fields_ = [("id", C.c_int), ("line", C.c_double * 6), ("init_kf_id", C.c_int), ("obs", C.POINTER(Obs)), ("num_obs", C.c_int)]

    class Packed(C.Structure):
        _fields_ = [("num_cameras", C.c_int), ("num_lines", C.c_int), ("num_observations", C.c_int), ("num_parameters", C.c_int),
                    ("camera_index", C.POINTER(C.c_int)), ("line_index", C.POINTER(C.c_int)), ("fixed_index", C.POINTER(C.c_int)),
                    ("observations", C.POINTER(C.c_double)), ("parameters", C.POINTER(C.c_double)),
                    ("camera_kf_id", C.POINTER(C.c_int)), ("line_lm_id", C.POINTER(C.c_int))]
    nk, nf, L = w["num_cameras"], w["num_free_cameras"], w["num_lines"]
    prm = w["parameters"]
    # camera slot c of the synthetic window is keyframe id: free slots 0..nf-1 -> ids nk-nf..nk-1, fixed -> 0..nk-nf-1
    slot_to_kf = list(range(nk - nf, nk)) + list(range(0, nk - nf))
    kf_to_slot = {k: s for s, k in enumerate(slot_to_kf)}
    keep = []
    kfs = (KF * nk)()
    members = {k: [] for k in range(nk)}
    for i, (c, l) in enumerate(zip(w["camera_index"], w["line_index"])):
        members[slot_to_kf[c]].append(int(l))
    for k in range(nk):
        kfs[k].id = k
        kfs[k].ba_rank = nk - 1 - k                     # newest keyframe has rank 0; rank < W <=> free
        wt = np.ascontiguousarray(prm[6 * kf_to_slot[k]:6 * kf_to_slot[k] + 6])
        lib.slslam_gc_wt_to_Rt(wt.ctypes.data_as(C.POINTER(C.c_double)), C.byref(kfs[k].T))
        arr = (C.c_int * max(1, len(members[k])))(*sorted(set(members[k])))
        keep.append(arr)
        kfs[k].member_lms = arr
        kfs[k].num_member_lms = len(set(members[k]))
    lms = (LM * L)()
    for l in range(L):
        sel = np.nonzero(w["line_index"] == l)[0]
        sel = sel[np.argsort([slot_to_kf[w["camera_index"][i]] for i in sel], kind="stable")]   # obs_vec is in time order
        obs = (Obs * len(sel))()
        for j, i in enumerate(sel):
            obs[j].kf_id = slot_to_kf[w["camera_index"][i]]
            obs[j].obs[:] = list(w["observations"][i])
        keep.append(obs)
        lms[l].id = 1000 + l
        lms[l].init_kf_id = obs[0].kf_id
        line_w = np.ascontiguousarray(synth.orth_to_av(prm[6 * nk + 4 * l:6 * nk + 4 * l + 4]))
        lc = np.zeros(6)
        lib.slslam_gc_line_to_pose(line_w.ctypes.data_as(C.POINTER(C.c_double)), C.byref(kfs[lms[l].init_kf_id].T),
                                   lc.ctypes.data_as(C.POINTER(C.c_double)))
        lms[l].line[:] = list(lc)
        lms[l].obs = obs
        lms[l].num_obs = len(sel)
    for k in range(nk):                                  # member ids must be landmark ids
        arr = (C.c_int * max(1, len(members[k])))(*[1000 + x for x in sorted(set(members[k]))])
        keep.append(arr)
        kfs[k].member_lms = arr
    return kfs, lms, Packed, keep, slot_to_kf


def test_se3_templates_of_the_mirrored_header(tmp_path):
    """gc_T_inv<T>, gc_w_20<T>, gc_T_20<T> (reference src/po_problem.h:27-64) are part of the header surface a caller of
    po_problem.h sees: slslam_amd/host/po_problem.h keeps them, on the four rotation helpers of host/ceres/rotation.h (restated
    from the published Ceres 1.7.0 definitions).  Instantiated for double and compared with the matrix forms of gc_lite
    (slslam_gc_T_inv, slslam_gc_T_20) over 4000 random pose pairs with rotation angles 0, 1e-9 ... pi."""
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = str(tmp_path / "se3_templates")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", "-Werror", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "host_cxx", "se3_templates.cpp"),
                           "-L", LIBDIR, "-lslslam_host", "-Wl,-rpath," + LIBDIR])
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    worst_inv, worst_comp, worst_quat = (float(v) for v in out)
    assert worst_inv < 1e-9 and worst_comp < 1e-12 and worst_quat < 1e-12, out


def test_ceres_harness_functor_known_answer(tmp_path):
    """tools/ceres_harness.cpp is the optional true-Ceres leg of the CPU baseline; it can only be built where a
    Ceres installation exists (bench.py::ceres_probe), which no box seen so far has.  What CAN be checked here is the residual it would hand to
    Ceres: its functor (tools/ceres_harness_functor.h), instantiated for double on the repo's own rotation helpers, reproduces the survey's
    known-answer vector at a general keyframe and at the identity keyframe."""
    exe = str(tmp_path / "ceres_functor_kat")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", "-Werror", "-I", HOST, "-I", os.path.join(ROOT, "tools"), "-o", exe,
                           os.path.join(ROOT, "tests", "host_cxx", "ceres_functor_kat.cpp")])
    worst = float(subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()[0])
    assert worst < 1e-14, worst


def test_window_packer_reproduces_the_array_contract(oracle):
    """slslam_pack_window (SLAM::bundle_adjustment pre, slam.cpp:811-921) on a map built from a synthetic
    window gives the same problem (same solve), and slslam_unpack_window (slam.cpp:957-972) writes the
    result back into the map."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", HOST])
    lib = C.CDLL(HOST_LIB)
    w = synth.make_window(8, num_lines=60, num_kf=12, num_free=5)
    kfs, lms, Packed, keep, slot_to_kf = _map_from_window(w, lib, C)
    pk = Packed()
    assert lib.slslam_pack_window(kfs, len(kfs), lms, len(lms), 5, C.byref(pk)) == 0
    Cn, L, M = pk.num_cameras, pk.num_lines, pk.num_observations
    assert (Cn, L, M) == (w["num_cameras"], w["num_lines"], len(w["camera_index"]))
    cam_kf = [pk.camera_kf_id[c] for c in range(Cn)]
    assert cam_kf[:5] == slot_to_kf[:5]                                  # free keyframes in ascending id
    assert sorted(cam_kf[5:]) == sorted(slot_to_kf[5:])
    w2 = dict(num_cameras=Cn, num_lines=L,
              camera_index=np.array([pk.camera_index[i] for i in range(M)]), line_index=np.array([pk.line_index[i] for i in range(M)]),
              fixed_index=np.array([pk.fixed_index[i] for i in range(2 * M)]),
              observations=np.array([pk.observations[i] for i in range(8 * M)]).reshape(M, 8),
              parameters=np.array([pk.parameters[i] for i in range(pk.num_parameters)]))
    assert np.all(w2["fixed_index"][1::2] == 0)
    assert np.array_equal(w2["fixed_index"][0::2], (w2["camera_index"] >= 5).astype(int))
    # same problem up to the camera-slot permutation of the constant keyframes: identical optimum
    x1, s1, _ = oracle.lba_solve(w, linear_solver=1)
    x2, s2, _ = oracle.lba_solve(w2, linear_solver=1)
    assert abs(s1["initial_cost"] - s2["initial_cost"]) < 1e-9 * s1["initial_cost"]
    assert abs(s1["final_cost"] - s2["final_cost"]) < 1e-7 * s1["final_cost"]
    slot_of = {k: s for s, k in enumerate(slot_to_kf)}
    for c in range(Cn):
        assert np.abs(x2[6 * c:6 * c + 6] - x1[6 * slot_of[cam_kf[c]]:6 * slot_of[cam_kf[c]] + 6]).max() < 1e-6
    # write-back
    for i in range(pk.num_parameters):
        pk.parameters[i] = x2[i]
    assert lib.slslam_unpack_window(C.byref(pk), kfs, len(kfs), lms, len(lms)) == 0
    back = np.zeros(6)
    lib.slslam_gc_Rt_to_wt(C.byref(kfs[cam_kf[0]].T), back.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(back - x2[:6]).max() < 1e-9
    lw = np.zeros(6)
    lib.slslam_gc_line_from_pose((C.c_double * 6)(*lms[0].line), C.byref(kfs[lms[0].init_kf_id].T), lw.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(lw - synth.orth_to_av(x2[6 * Cn:6 * Cn + 4])).max() < 1e-9
    lib.slslam_free_packed_window(C.byref(pk))


def _host_lib():
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", HOST])
    return C, C.CDLL(HOST_LIB)


def test_trajectory_writer_reproduces_the_reference_files():
    """Lines of trajectory files the reference ships (output of SLAM::save_trajectory, slam.cpp:1470-1496):
    rebuild the keyframe pose each line encodes, format it with slslam_format_trajectory_line -> same text."""
    C, lib = _host_lib()

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]
    golden = open(os.path.join(ROOT, "tests", "golden", "traj_excerpt.txt")).read().splitlines(True)
    assert len(golden) == 25
    buf = C.create_string_buffer(512)
    for ln in golden:
        f = ln.split("\t")
        idx, z, mx, my, w = int(f[0]), float(f[1]), float(f[2]), float(f[3]), np.array([float(v) for v in f[4:7]])
        Ti = Pose()                                           # gc_T_inv(kf->T): camera position / orientation in the root frame
        lib.slslam_gc_rodrigues_to_R((C.c_double * 3)(*w), Ti.R)
        Ti.t[:] = [-mx + 0.0, -my + 0.0, z]
        T = Pose()
        lib.slslam_gc_T_inv(C.byref(Ti), C.byref(T))
        assert lib.slslam_format_trajectory_line(idx, C.byref(T), buf, 512) == 0
        got = buf.value.decode()
        if idx == 0:                                          # the root line: signs of zero are not recoverable from text
            assert got.replace("-0", "0") == ln.replace("-0", "0")
        else:
            assert got == ln


def test_trajectory_writer_reproduces_every_shipped_trajectory_file():
    """Build-container check: ALL lines of the six trajectory files the reference ships (matlab_script/traj_slslam_*.txt, 1354
    keyframes of the it3f / myungdong / olympic4f runs) go text -> pose -> slslam_format_trajectory_line -> the same text.
    Reads the reference tree at test time (nothing of it is committed beyond the 25-line excerpt above); skipped where absent."""
    import glob
    files = sorted(glob.glob("/root/reference/matlab_script/traj_slslam_*.txt"))
    if not files:
        pytest.skip("reference tree not present")
    C, lib = _host_lib()

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]
    buf = C.create_string_buffer(512)
    total = mismatched = 0
    for fn in files:
        for ln in open(fn).read().splitlines(True):
            f = ln.split("\t")
            if len(f) != 7:
                continue
            idx, z, mx, my, w = int(f[0]), float(f[1]), float(f[2]), float(f[3]), np.array([float(v) for v in f[4:7]])
            Ti = Pose()
            lib.slslam_gc_rodrigues_to_R((C.c_double * 3)(*w), Ti.R)
            Ti.t[:] = [-mx + 0.0, -my + 0.0, z]
            T = Pose()
            lib.slslam_gc_T_inv(C.byref(Ti), C.byref(T))
            assert lib.slslam_format_trajectory_line(idx, C.byref(T), buf, 512) == 0
            got = buf.value.decode()
            total += 1
            # six significant digits in the file: re-deriving the pose from the printed text and printing it again may move the
            # last digit of a value that sat on a rounding boundary; anything else is a format difference
            if got.replace("-0\t", "0\t") != ln.replace("-0\t", "0\t"):
                a, b = got.split("\t"), ln.split("\t")
                assert len(a) == len(b) and a[0] == b[0]
                for x, y in zip(a[1:], b[1:]):
                    assert abs(float(x) - float(y)) <= 2e-6 * max(abs(float(y)), 1e-3), (fn, ln, got)
                mismatched += 1
    assert total > 1300 and mismatched == 0, (total, mismatched)      # measured: all 1354 lines byte for byte


def test_frame_reader_and_metric_embedding(tmp_path):
    C, lib = _host_lib()

    class Intr(C.Structure):
        _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double)]

    class Frame(C.Structure):
        _fields_ = [("num_lines", C.c_int), ("ids", C.POINTER(C.c_int)), ("observations", C.POINTER(C.c_double))]

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    class Edge(C.Structure):
        _fields_ = [("from_", C.c_int), ("to", C.c_int), ("T", Pose)]
    # ---- reader: "id x0 y0 x1 y1 x2 y2 x3 y3 extra", unsorted ids, a duplicate, an alias, a blank line
    K = Intr(406.05, 406.05, 327.783, 237.172)
    rows = {17: [100.5, 50.25, 200, 60, 90, 50, 190, 60], 3: [10, 20, 30, 40, 5, 20, 25, 40], 9: [1, 2, 3, 4, 5, 6, 7, 8]}
    text = "17 " + " ".join(str(v) for v in rows[17]) + " 0.5\n" + "3 " + " ".join(str(v) for v in rows[3]) + "\n\n" + \
           "17 0 0 0 0 0 0 0 0\n" + "9 " + " ".join(str(v) for v in rows[9]) + " 1\n"
    path = tmp_path / "0007.txt"
    path.write_text(text)
    buf = C.create_string_buffer(512)
    assert lib.slslam_frame_path(str(tmp_path).encode(), 7, buf, 512) == 0 and buf.value.decode() == str(path)
    fr = Frame()
    af, at = (C.c_int * 1)(9), (C.c_int * 1)(2)               # match_lookup: feature 9 is landmark 2
    assert lib.slslam_read_frame_file(str(path).encode(), C.byref(K), af, at, 1, C.byref(fr)) == 0
    assert [fr.ids[i] for i in range(fr.num_lines)] == [2, 3, 17]          # std::map order, first duplicate wins
    want = {2: rows[9], 3: rows[3], 17: rows[17]}
    for i in range(fr.num_lines):
        px = np.array(want[fr.ids[i]])
        c = np.array([327.783, 237.172] * 4)
        got = np.array([fr.observations[8 * i + q] for q in range(8)])
        assert np.abs(got - (px / 406.05 - c / 406.05)).max() < 1e-15       # obs / f - c / f (slam.cpp:121-128)
    lib.slslam_free_frame(C.byref(fr))
    assert lib.slslam_read_frame_file(b"/nonexistent/0001.txt", C.byref(K), None, None, 0, C.byref(fr)) == 1
    # ---- metric embedding on a chain 0-1-2-3 with a shortcut 0-3: poses compose along the graph walk
    rng = np.random.default_rng(5)
    true = [np.r_[rng.normal(size=3) * 0.1, rng.normal(size=3)] for _ in range(4)]
    true[0][:] = 0
    P = (Pose * 4)()
    for i in range(4):
        lib.slslam_gc_wt_to_Rt((C.c_double * 6)(*true[i]), C.byref(P[i]))
    pairs = [(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2), (0, 3), (3, 0)]
    E = (Edge * len(pairs))()
    for e, (a, b) in enumerate(pairs):
        E[e].from_, E[e].to = a, b
        lib.slslam_gc_T_21(C.byref(P[b]), C.byref(P[a]), C.byref(E[e].T))   # T_{b<-a}
    ids = (C.c_int * 4)(0, 1, 2, 3)
    nbr_ptr = (C.c_int * 5)(0, 2, 4, 6, 8)
    nbr = (C.c_int * 8)(1, 3, 0, 2, 1, 3, 0, 2)
    out = (Pose * 4)()
    order, dist, n = (C.c_int * 4)(), (C.c_double * 4)(), C.c_int(0)
    assert lib.slslam_metric_embedding(0, 4, ids, nbr_ptr, nbr, E, len(pairs), out, order, dist, C.byref(n)) == 0
    assert n.value == 4 and order[0] == 0 and dist[0] == 0.0 and list(dist) == sorted(dist)
    for i in range(4):
        wt = np.zeros(6)
        lib.slslam_gc_Rt_to_wt(C.byref(out[i]), wt.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(wt - true[i]).max() < 1e-12
    assert lib.slslam_metric_embedding(42, 4, ids, nbr_ptr, nbr, E, len(pairs), out, order, dist, C.byref(n)) == 1
    # ---- landmark endpoints: a segment given by (point, direction) + end parameters in its keyframe
    line = np.r_[[0.3, -0.2, 4.0], [0.6, 0.0, 0.8]]
    ep = np.zeros(6)
    lib.slslam_landmark_endpoints(line.ctypes.data_as(C.POINTER(C.c_double)), (C.c_double * 2)(-0.5, 1.5), C.byref(P[2]),
                                  ep.ctypes.data_as(C.POINTER(C.c_double)))
    v = line[3:] / np.linalg.norm(line[3:])
    p0 = line[:3] - v * (line[:3] @ v)
    R2 = np.array(P[2].R).reshape(3, 3); t2 = np.array(P[2].t)
    for e, s in enumerate((-0.5, 1.5)):
        assert np.abs(ep[3 * e:3 * e + 3] - R2.T @ (p0 + v * s - t2)).max() < 1e-12


def test_pose_graph_packer_reproduces_the_array_contract(oracle):
    """slslam_pack_pose_graph (SLAM::pose_optimization pre, slam.cpp:1248-1280) on a keyframe / edge map built from a
    synthetic pose graph gives the arrays of the POProblem contract - edges in std::set<pii> order whatever order they
    were handed over in - and slslam_unpack_pose_graph (slam.cpp:1295-1311) writes the solved poses back and refreshes
    the edges' relative poses."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", HOST])
    lib = C.CDLL(HOST_LIB)

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    class Edge(C.Structure):
        _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("C", Pose), ("T", Pose), ("T_rev", Pose)]

    class Packed(C.Structure):
        _fields_ = [("num_poses", C.c_int), ("num_edges", C.c_int), ("pose_index_1", C.POINTER(C.c_int)),
                    ("pose_index_2", C.POINTER(C.c_int)), ("constraints", C.POINTER(C.c_double)), ("parameters", C.POINTER(C.c_double))]

    def pose_of(wt):
        T = Pose()
        lib.slslam_gc_wt_to_Rt((C.c_double * 6)(*wt), C.byref(T))
        return T

    g = synth.make_pose_graph(4, num_poses=30, num_loops=3)
    N, E = g["num_poses"], len(g["pose_index_1"])
    kfT = (Pose * N)(*[pose_of(g["parameters"][6 * k:6 * k + 6]) for k in range(N)])
    perm = np.random.default_rng(0).permutation(E)                      # the packer restores the set order
    edges = (Edge * E)()
    for slot, e in enumerate(perm):
        edges[slot].n1, edges[slot].n2 = int(g["pose_index_1"][e]), int(g["pose_index_2"][e])
        edges[slot].C = pose_of(g["constraints"][e])
    pk = Packed()
    assert lib.slslam_pack_pose_graph(kfT, N, edges, E, C.byref(pk)) == 0
    assert (pk.num_poses, pk.num_edges) == (N, E)
    assert [pk.pose_index_1[i] for i in range(E)] == list(g["pose_index_1"]) and [pk.pose_index_2[i] for i in range(E)] == list(g["pose_index_2"])
    cons = np.array([pk.constraints[i] for i in range(6 * E)]).reshape(E, 6)
    prm = np.array([pk.parameters[i] for i in range(6 * N)])
    assert np.abs(cons - g["constraints"]).max() < 1e-12 and np.abs(prm - g["parameters"]).max() < 1e-12
    # solve (oracle on the CPU) and write back
    x, s, _ = oracle.po_solve(dict(g, pose_index_1=np.array([pk.pose_index_1[i] for i in range(E)], dtype=np.int32),
                                   pose_index_2=np.array([pk.pose_index_2[i] for i in range(E)], dtype=np.int32),
                                   constraints=cons, parameters=prm))
    for i in range(6 * N):
        pk.parameters[i] = x[i]
    assert lib.slslam_unpack_pose_graph(C.byref(pk), kfT, N, edges, E) == 0
    back = np.zeros(6)
    lib.slslam_gc_Rt_to_wt(C.byref(kfT[N - 1]), back.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(back - x[6 * (N - 1):]).max() < 1e-9
    # after the optimisation every edge's current relative pose is close to its measurement, in both directions
    for e in range(E):
        rel, rev = np.zeros(6), np.zeros(6)
        lib.slslam_gc_Rt_to_wt(C.byref(edges[e].T), rel.ctypes.data_as(C.POINTER(C.c_double)))
        lib.slslam_gc_Rt_to_wt(C.byref(edges[e].T_rev), rev.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(rel - cons[e]).max() < 5e-2
        inv = Pose()
        lib.slslam_gc_T_inv(C.byref(edges[e].T), C.byref(inv))
        chk = np.zeros(6)
        lib.slslam_gc_Rt_to_wt(C.byref(inv), chk.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(chk - rev).max() < 1e-9
    # an edge naming an unknown pose and a duplicate edge are rejected
    bad = (Edge * 1)(); bad[0].n1, bad[0].n2 = 0, N
    assert lib.slslam_pack_pose_graph(kfT, N, bad, 1, C.byref(Packed())) == 1
    dup = (Edge * 2)(); dup[0].n1, dup[0].n2, dup[1].n1, dup[1].n2 = 0, 1, 0, 1
    assert lib.slslam_pack_pose_graph(kfT, N, dup, 2, C.byref(Packed())) == 1
    lib.slslam_free_packed_pose_graph(C.byref(pk))


def test_motion_only_packer_reproduces_the_array_contract(oracle):
    """slslam_pack_motion_only (SLAM::motion_only_ba pre, slam.cpp:590-640): the arrays it emits are those of the
    synthetic motion-only window they were taken apart from; slslam_unpack_motion_only returns camera 0."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", HOST])
    lib = C.CDLL(HOST_LIB)

    class Pose(C.Structure):
        _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]

    class Packed(C.Structure):
        _fields_ = [("num_cameras", C.c_int), ("num_lines", C.c_int), ("num_observations", C.c_int), ("num_parameters", C.c_int),
                    ("camera_index", C.POINTER(C.c_int)), ("line_index", C.POINTER(C.c_int)), ("fixed_index", C.POINTER(C.c_int)),
                    ("observations", C.POINTER(C.c_double)), ("parameters", C.POINTER(C.c_double)),
                    ("camera_kf_id", C.POINTER(C.c_int)), ("line_lm_id", C.POINTER(C.c_int))]

    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    w = synth.make_motion_only(3, num_lines=40)
    K = w["num_lines"]
    prm = w["parameters"]
    assert np.abs(prm[6:12]).max() == 0.0                                  # camera 1 is the identity keyframe
    obs = w["observations"]
    cur = np.zeros((K, 8)); prev = np.zeros((K, 8)); have = np.zeros((K, 2), dtype=bool)
    for i in range(len(w["camera_index"])):
        (cur if w["camera_index"][i] == 0 else prev)[w["line_index"][i]] = obs[i]
        have[w["line_index"][i], w["camera_index"][i]] = True
    sel = np.nonzero(have.all(axis=1))[0]                                  # lines seen in both frames = the inliers
    lines_av = np.array([synth.orth_to_av(prm[12 + 4 * l:16 + 4 * l]) for l in sel])
    T = Pose()
    lib.slslam_gc_wt_to_Rt((C.c_double * 6)(*prm[:6]), C.byref(T))
    pk = Packed()
    cur_s, prev_s = np.ascontiguousarray(cur[sel]), np.ascontiguousarray(prev[sel])
    assert lib.slslam_pack_motion_only(C.byref(T), dp(cur_s), dp(prev_s), dp(np.ascontiguousarray(lines_av)), len(sel), C.byref(pk)) == 0
    n = len(sel)
    assert (pk.num_cameras, pk.num_lines, pk.num_observations, pk.num_parameters) == (2, n, 2 * n, 12 + 4 * n)
    assert [pk.camera_index[i] for i in range(2 * n)] == [0, 1] * n and [pk.line_index[i] for i in range(2 * n)] == [i // 2 for i in range(2 * n)]
    assert [pk.fixed_index[i] for i in range(4 * n)] == [0, 1, 1, 1] * n
    o = np.array([pk.observations[i] for i in range(16 * n)]).reshape(n, 2, 8)
    assert np.array_equal(o[:, 0], cur_s) and np.array_equal(o[:, 1], prev_s)
    p = np.array([pk.parameters[i] for i in range(12 + 4 * n)])
    assert np.abs(p[:6] - prm[:6]).max() < 1e-12 and np.abs(p[6:12]).max() == 0.0
    # the orthonormal line parameters come back the same up to the representation's sign / 2 pi freedom: same lines
    for q, l in enumerate(sel):
        assert np.abs(synth.orth_to_av(p[12 + 4 * q:16 + 4 * q]) - lines_av[q]).max() < 1e-9
    # the packed problem is the generator's (restricted to the inliers): same optimum for camera 0
    w2 = dict(num_cameras=2, num_lines=n, camera_index=np.array([0, 1] * n, dtype=np.int32), line_index=np.repeat(np.arange(n, dtype=np.int32), 2),
              fixed_index=np.array([0, 1, 1, 1] * n, dtype=np.int32), observations=o.reshape(2 * n, 8), parameters=p)
    x2, s2, _ = oracle.lba_solve(w2, linear_solver=1)
    x1, s1, _ = oracle.lba_solve(w, linear_solver=1)
    if n == K:
        assert abs(s1["final_cost"] - s2["final_cost"]) < 1e-7 * s1["final_cost"] and np.abs(x1[:6] - x2[:6]).max() < 1e-7
    for i in range(6):
        pk.parameters[i] = x2[i]
    lib.slslam_unpack_motion_only(C.byref(pk), C.byref(T))
    back = np.zeros(6)
    lib.slslam_gc_Rt_to_wt(C.byref(T), dp(back))
    assert np.abs(back - x2[:6]).max() < 1e-9
    lib.slslam_free_packed_window(C.byref(pk))


@pytest.mark.gpu
def test_house_replay_cxx_equals_python(tmp_path, hip):
    """The host library composed end to end in C++ (tests/host_cxx/house_replay.cpp: per keyframe slslam_pack_motion_only ->
    slslam_lba_solve -> unpack, slslam_pack_window -> slslam_lba_solve -> slslam_unpack_window on the reference's map structures,
    then the trajectory writer) against the numpy restatement of the same data flow (tools/house_study.py::run_reference_protocol)
    on the simulated house run: the same solver, the glue written twice.  200 keyframes, W = 10, sigma = 0.2 px: 199 window
    solves and 199 motion-only solves in a closed loop.
      * exact: what a window takes from the map's bookkeeping - camera / line / constness indices and the observations of every
        one of the 199 windows (FNV digests);
      * the trajectory relative to keyframe 0 (what save_trajectory writes) to 1e-9 (measured 5e-14): the two glue codes differ in
        the last bits only.  (The loop is sensitive where the reference's own pipeline is: the first W - 1 windows have no constant
        keyframe, so the absolute poses carry a gauge offset of up to 1e-6 between the two, and on noisier / smaller-window runs
        a single accept / reject decision that flips sends the two runs apart - neither is a property of the glue.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import house_study
    subprocess.check_call(["make", "-s", "-C", HOST])
    exe = os.path.join(ROOT, "tests", "_build", "house_replay")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "host_cxx", "house_replay.cpp"),
                           "-L", LIBDIR, "-lslslam_host", "-lslslam_hip", "-Wl,-rpath," + LIBDIR])
    scene = str(tmp_path / "scene.bin")
    N = 200
    py = house_study.run_reference_protocol(0.2, 10, lambda w, it: hip.lba_solve(w, max_num_iterations=it)[:2], frames=N, dump=scene)
    subprocess.check_call([exe, scene, str(tmp_path / "poses.bin"), str(tmp_path / "traj.txt")])
    raw = np.fromfile(tmp_path / "poses.bin")
    poses, tail = raw[:12 * N].reshape(N, 12), raw[12 * N:12 * N + 3]
    dig = np.fromfile(tmp_path / "poses.bin", dtype=np.uint64)[12 * N + 3:].reshape(-1, 4)
    assert py["lba_calls"] == N - 1 and py["motion_only_calls"] >= N - 10
    assert len(dig) == len(py["window_digests"]) == N - 1
    for got, want in zip(dig, py["window_digests"]):
        assert tuple(int(v) for v in got) == tuple(int(v) for v in want)
    assert int(tail[0]) == py["lm_iterations"]                                    # every solve took the same number of LM steps
    assert abs(tail[1] - py["sum_initial_cost"]) <= 1e-8 * py["sum_initial_cost"]
    assert abs(tail[2] - py["sum_final_cost"]) <= 1e-8 * py["sum_final_cost"]

    def reroot(P):
        R0, t0 = P[0, :9].reshape(3, 3), P[0, 9:]
        out = []
        for q in P:
            Rr = q[:9].reshape(3, 3) @ R0.T
            out.append(np.concatenate([Rr.reshape(-1), q[9:] - Rr @ t0]))
        return np.array(out)
    rel_c, rel_p = reroot(poses), reroot(py["poses"])
    assert np.abs(rel_c - rel_p).max() < 1e-9
    assert np.abs(poses - py["poses"]).max() < 1e-5
    assert np.linalg.norm(rel_c[N // 2, 9:]) > 1.0                                 # half way round the house
    # the trajectory file: one line per keyframe, "i z -x -y w0 w1 w2" of the camera pose relative to keyframe 0 (save_trajectory)
    lines = open(tmp_path / "traj.txt").read().splitlines()
    assert len(lines) == N
    for k in (0, 37, N - 1):
        f = lines[k].split("\t")
        Rr, tr = rel_c[k, :9].reshape(3, 3), rel_c[k, 9:]
        c = -(Rr.T @ tr)                                                              # gc_T_inv: camera position in the root frame
        assert int(f[0]) == k
        assert np.allclose([float(f[1]), float(f[2]), float(f[3])], [c[2], -c[0], -c[1]], rtol=2e-5, atol=2e-6)
