#!/bin/bash
# kept-Jacobian sweep: parity tests of the grouped sweep, then the short bench with lba_keep_jacobian = 1 / 0 on the same box.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-keep}
timeout 1500 python -m pytest tests/test_gpu_lba.py -x -q -k "kept_jacobian or matrix_core or grouped" > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
: > gpurun_out/${TAG}_ab.txt
for KEEP in ${KEEPS:-0 1 0}; do
  timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 --keep-jacobian $KEEP ${BENCH_ARGS} --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_b${KEEP}.json 2> gpurun_out/${TAG}_b${KEEP}.err
  python - $KEEP gpurun_out/${TAG}_b${KEEP}.json >> gpurun_out/${TAG}_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    k=d["kernel_ms_per_step"]
    print("keep %s  value %.0f  ms/step %.3f  K1 %.4f  backsub %.4f  solve %.4f ms/launch  check %s  consistency %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["reduced_solve_mfma"]["avg_launch_ms"], (d.get("results_check") or {}).get("bitwise_equal_to_rank0_resolve"), d.get("launch_consistency")))
except Exception as e:
    print("keep %s FAILED %r" % (sys.argv[1], e))
PY
done
cat gpurun_out/${TAG}_ab.txt
