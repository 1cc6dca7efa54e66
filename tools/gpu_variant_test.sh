#!/bin/bash
# builds the library with extra flags on the GPU box, runs the LBA GPU tests and the short bench, restores the default build
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
FLAGS="$1"
SLSLAM_EXTRA_FLAGS="$FLAGS" python -c "from slslam_amd import build; build.build_lib(force=True)" > gpurun_out/vt_build.log 2>&1 || { echo BUILD FAILED; tail -5 gpurun_out/vt_build.log; }
python -m pytest tests/test_gpu_lba.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/vt_b.json 2> gpurun_out/vt_b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/vt_b.json"))
print("value %.0f K1 %.4f backsub %.4f check %s" % (d["value"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["results_check"]["bitwise_equal_to_rank0_resolve"]))
PY
python -c "from slslam_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
