"""Host-side code and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 "Race detection /
sanitizers"; zero GPU minutes).  What is instrumented (-fsanitize=address,undefined -fno-sanitize-recover=all):

  * slslam_amd/csrc/lba_pack.cpp + the math / index-map headers   through tests/host_math (SLSLAM_SANITIZE=1)
  * slslam_amd/host/{problems,gc_lite,window_packer,sequence_io}.cpp   `make -C slslam_amd/host asan` (SLSLAM_HOST_LIB)
  * oracle/*.c                                                    `make -C oracle asan` (SLSLAM_ORACLE_LIB)

and what runs against them: the CPU tests of those layers (tests/test_host_side.py: packer invariants incl. oversize windows, long
and short lines, edge cases, the index-map replays; tests/test_oracle.py; tests/test_host_cxx.py: boundary encodings, the three
packers, sequence I/O) in a python that has libasan preloaded, and the reference's ownership protocol (the problem object
delete[]s the caller's new[] arrays: src/lba_problem.cpp:46-52, src/po_problem.cpp:33-38) as a stand-alone instrumented executable
with leak detection on.  Round 2's advisor found exactly this class of bug by reading (a shift past the width of int in the packer);
any report fails the test."""
import os
import subprocess
import sys

import numpy as np
import pytest

from slslam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slslam_amd", "host")
LIBDIR = os.path.join(ROOT, "slslam_amd", "_lib")
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not p or not os.path.exists(p):
        pytest.skip("gcc has no libasan here")
    return os.path.realpath(p)


def _build_sanitized():
    from slslam_amd import build
    build.build_lib()                                     # the host library links against the HIP library's C ABI
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    subprocess.check_call(["make", "-s", "-C", HOST, "asan"])
    return os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so"), os.path.join(LIBDIR, "libslslam_host_asan.so")


def test_cpu_suites_of_packer_host_library_and_oracle_under_asan_ubsan():
    libasan = _libasan()
    oracle_lib, host_lib = _build_sanitized()
    env = dict(os.environ, LD_PRELOAD=libasan, SLSLAM_SANITIZE="1", SLSLAM_ORACLE_LIB=oracle_lib, SLSLAM_HOST_LIB=host_lib,
               # CPython itself leaks at exit and is not instrumented: errors, not leaks, are what this leg looks for
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    # (left out: the tests that compile the HIP library with hipcc or run stand-alone executables - neither is instrumented)
    skip = "not keep_their_occupancy and not compiles_and_fails and not house_replay and not drop_in and not se3_templates"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_host_side.py", "tests/test_oracle.py", "tests/test_host_cxx.py", "-k", skip],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail


def _write_lba(path, w, max_iter=10, robust=1):
    hdr = np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"]), max_iter, robust], dtype=np.int32)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        for k, dt in (("camera_index", np.int32), ("line_index", np.int32), ("fixed_index", np.int32),
                      ("observations", np.float64), ("parameters", np.float64)):
            f.write(np.ascontiguousarray(w[k], dtype=dt).tobytes())


def test_reference_ownership_protocol_under_asan_with_leak_detection(tmp_path):
    """tests/host_cxx/drop_in_demo.cpp (the reference's call protocol re-typed: new[] the arrays, hand them to the problem object,
    build, set_options, ceres::Solve, read `parameters`, let the object die) built with the sanitizers against the sanitized host
    library.  Without a GPU the solve returns SLSLAM_ERR_NO_DEVICE - the whole ownership path (setters, marshalling, the
    destructor's delete[] of all five / four arrays) still runs.  ASan checks new[] / delete[] pairing and use after free,
    LeakSanitizer that nothing the problem object owns survives it (leaks inside the uninstrumented HIP runtime are suppressed)."""
    _libasan()
    _, host_lib = _build_sanitized()
    from slslam_amd import capi
    exe = str(tmp_path / "drop_in_demo_asan")
    subprocess.check_call(["g++"] + SAN + ["-std=c++11", "-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "host_cxx", "drop_in_demo.cpp"),
                           host_lib, "-L", LIBDIR, "-lslslam_hip", "-Wl,-rpath," + LIBDIR])
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:libamd_comgr\nleak:librocprofiler\nleak:libslslam_hip\nleak:hipGetDeviceCount\nleak:dl_init\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:alloc_dealloc_mismatch=1:new_delete_type_mismatch=1",
               LSAN_OPTIONS="suppressions=%s:print_suppressions=0" % supp, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    have_gpu = capi.device_count() > 0
    for seed, kw in ((3, dict(num_lines=20)), (4, dict(num_lines=60, num_kf=8, num_free=3))):
        w = synth.make_window(seed, **kw)
        _write_lba(tmp_path / "in.bin", w)
        p = subprocess.run([exe, "lba", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, env=env, timeout=600)
        assert "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
        assert p.returncode == (0 if have_gpu else 2), p.stderr[-2000:]       # 2 = SLSLAM_ERR_NO_DEVICE, reported by the demo
    g = synth.make_pose_graph(5, num_poses=30, num_loops=2)
    with open(tmp_path / "po.bin", "wb") as f:
        np.array([g["num_poses"], len(g["pose_index_1"])], dtype=np.int32).tofile(f)
        for k, dt in (("pose_index_1", np.int32), ("pose_index_2", np.int32), ("constraints", np.float64), ("parameters", np.float64)):
            np.asarray(g[k], dtype=dt).tofile(f)
    p = subprocess.run([exe, "po", str(tmp_path / "po.bin"), str(tmp_path / "po_out.bin")], capture_output=True, text=True, env=env, timeout=600)
    assert "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == (0 if have_gpu else 2), p.stderr[-2000:]
