#!/bin/bash
# the short bench through variant builds of the library (tools/variant_lib.sh), interleaved with the product build on the same box
#   tools/gpu_variant_compare.sh <tag> <variant> [variant ...]
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
: > gpurun_out/${TAG}.txt
for rep in 1 2; do
for V in product "$@"; do
  if [ $V = product ]; then unset SLSLAM_HIP_LIBRARY; else export SLSLAM_HIP_LIBRARY=/root/repo/slslam_amd/_lib/variants/$V.so; fi
  timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  python - $V gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rc=d.get("results_check") or {}
    print("%-12s value %8.0f  ms/step %7.3f  K1 %.4f  backsub %.4f  solve %.4f  digest %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"], (rc.get("equal_to_stored_1_rank_digest") or {}).get("equal")))
except Exception as e:
    print("%s FAILED %r" % (sys.argv[1], e))
PY
done
done
cat gpurun_out/${TAG}.txt
