import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): compiled on demand with gcc.  Tests only."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def host_math():
    """The product's device math header compiled for the host (tests/host_math)."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "host_math", "host_math.cpp")
    out_dir = os.path.join(ROOT, "tests", "_build")
    san = bool(os.environ.get("SLSLAM_SANITIZE"))          # tests/test_sanitizers.py: the packer and the math headers under ASan + UBSan
    out = os.path.join(out_dir, "libhost_math_asan.so" if san else "libhost_math.so")
    deps = [src, os.path.join(ROOT, "slslam_amd", "csrc", "lba_math.h"), os.path.join(ROOT, "slslam_amd", "csrc", "lba_pack.cpp"),
            os.path.join(ROOT, "slslam_amd", "csrc", "lba_pack.h"), os.path.join(ROOT, "slslam_amd", "csrc", "lba_types.h"),
            os.path.join(ROOT, "slslam_amd", "csrc", "lba_eliminate_grouped_maps.h"), os.path.join(ROOT, "slslam_amd", "csrc", "lba_eliminate_mfma_maps.h"),
            os.path.join(ROOT, "slslam_amd", "csrc", "lba_gram.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(out_dir, exist_ok=True)
        flags = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"] if san else ["-O2"]
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-shared", "-fPIC", "-o", out, src,
                               os.path.join(ROOT, "slslam_amd", "csrc", "lba_pack.cpp")])
    return C.CDLL(out)


@pytest.fixture(scope="session")
def hip():
    """The product library through its ctypes binding; GPU tests fail loudly if it is missing."""
    from slslam_amd import capi
    capi.lib()
    if capi.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    return capi
