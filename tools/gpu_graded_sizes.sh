#!/bin/bash
# graded against equal chunk sizes by batch size (automatic sweep): LM it/s of the short bench
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-gs}
: > gpurun_out/${TAG}.txt
for W in ${SIZES:-256 512 768 1024 1536 2048}; do
  for EQ in 0 1; do
    if [ $EQ = 1 ]; then export SLSLAM_EQUAL_CHUNKS=1; else unset SLSLAM_EQUAL_CHUNKS; fi
    timeout 400 python bench.py --steps 10 --warmup 2 --windows $W --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
    python - $W $EQ gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3]))
    print("windows %5s %s: value %8.0f  ms/step %7.3f  K1 %.4f  backsub %.4f  solve %.4f  sweep %s  bitwise %s" % (sys.argv[1], "equal " if sys.argv[2]=="1" else "graded", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"], d["roofline"].get("lba_elimination"), (d.get("results_check") or {}).get("bitwise_equal_to_rank0_resolve")))
except Exception as e:
    print("windows %s eq %s FAILED %r" % (sys.argv[1], sys.argv[2], e))
PY
  done
done
cat gpurun_out/${TAG}.txt
