"""In-tree build of the HIP library (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB = os.path.join(LIB_DIR, "libslslam_hip.so")
SOURCES = ["lba_api.hip", "lba_pack.cpp", "po_api.hip", "ransac_api.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "slslam_hip.h")]   # every header: a new one must not leave a stale library behind
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics"]
import os as _os
FLAGS += _os.environ.get("SLSLAM_EXTRA_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_lib(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 into slslam_amd/_lib/libslslam_hip.so."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
