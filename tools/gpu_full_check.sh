#!/bin/bash
# the whole GPU test-suite, then the default bench line (what the driver runs)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-full}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_gputests.log
tail -n 6 gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value","ms_per_step","kernel_ms_per_step") if k in d})
print("roofline", d.get("roofline")); print("backsub", d.get("roofline_backsub")); print("traj", d.get("traj_error_vs_oracle")); print("results_check", d.get("results_check"))
PY
