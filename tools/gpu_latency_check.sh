#!/bin/bash
# GPU tests of the LBA path, then the latency entries of the bench line (one resident window at the reference's study sizes)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-lat}
timeout 1800 python -m pytest tests/test_gpu_lba.py tests/test_host_cxx.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.log
tail -n 4 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-run > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value", d["value"]); print("latency_single_window", d.get("latency_single_window")); print("latency_study_windows", d.get("latency_study_windows"))
PY
