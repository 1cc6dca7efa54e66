#!/bin/bash
# graded chunk sizes: throughput of the short bench by the weights of a window's chunks (comma-separated lists; "equal" = equal chunks; "default" = the library's)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-cg}; shift
: > gpurun_out/${TAG}.txt
for LT in "$@"; do
  unset SLSLAM_EQUAL_CHUNKS SLSLAM_CHUNK_WEIGHTS
  if [ "$LT" = "equal" ]; then export SLSLAM_EQUAL_CHUNKS=1; elif [ "$LT" != "default" ]; then export SLSLAM_CHUNK_WEIGHTS=$LT; fi
  timeout 300 python bench.py --steps 10 --warmup 2 ${BENCH_ARGS} --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  python - $LT gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("weights %-16s value %8.0f  ms/step %7.3f  K1 %.4f  backsub %.4f  solve %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"]))
except Exception as e:
    print("weights %s FAILED %r" % (sys.argv[1], e))
PY
done
cat gpurun_out/${TAG}.txt
