#!/bin/bash
# persistent sweeps against one workgroup per chunk: the short bench in both modes (same box, interleaved), bytes against the stored digest
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_persist}
: > gpurun_out/${TAG}.txt
for rep in 1 2; do
for MODE in off on; do
  if [ $MODE = on ]; then export SLSLAM_PERSISTENT_SWEEPS=1; else unset SLSLAM_PERSISTENT_SWEEPS; fi
  timeout 400 python bench.py --steps 10 --warmup 2 ${BENCH_ARGS} --no-cpu-baseline --no-overlap-run --no-extra-configs --no-streamed > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  python - $MODE gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rc=d.get("results_check") or {}
    print("persistent %-3s value %8.0f  ms/step %7.3f  K1 %.4f  backsub %.4f  solve %.4f  digest %s bitwise %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"], (rc.get("equal_to_stored_1_rank_digest") or {}).get("equal"), rc.get("bitwise_equal_to_rank0_resolve")))
except Exception as e:
    print("persistent %s FAILED %r" % (sys.argv[1], e)); print(open(sys.argv[2].replace('.json','.err')).read()[-800:])
PY
done
done
cat gpurun_out/${TAG}.txt
