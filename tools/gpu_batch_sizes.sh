#!/bin/bash
# throughput by batch size and elimination sweep (does the automatic choice pick the faster one?): LM it/s of the short bench
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-bs}
: > gpurun_out/${TAG}.txt
for W in ${SIZES:-16 64 128 256 512 1024}; do
  for E in 0 1 4; do
    timeout 300 python bench.py --steps 10 --warmup 2 --windows $W --elim $E --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
    python - $W $E gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3]))
    print("windows %5s elim %s -> %s: value %8.0f  ms/step %7.3f  K1 %.4f  backsub %.4f  solve %.4f" % (sys.argv[1], sys.argv[2], d["roofline"].get("lba_elimination"), d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"]))
except Exception as e:
    print("windows %s elim %s FAILED %r" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
cat gpurun_out/${TAG}.txt
