cd /root/repo; export TMPDIR=/tmp
SLSLAM_EXTRA_FLAGS="-DSLSLAM_K1_TIMING=1" python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1 || echo BUILD FAILED
python tools/k1_phases.py
python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
