#!/bin/bash
# resident and streamed throughput by windows per batch (final code of round 5): the short bench with its streamed leg
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_sizes}
: > gpurun_out/${TAG}.txt
for W in ${SIZES:-128 256 512 768 1024 1536 2048}; do
  timeout 600 python bench.py --steps 10 --warmup 2 --windows $W --stream-batches 24 --no-cpu-baseline --no-overlap-run --no-extra-configs --no-result-check > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  python - $W gpurun_out/${TAG}_b.json >> gpurun_out/${TAG}.txt <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); s=d.get("streamed") or {}
    print("windows %5s: resident %8.0f LM it/s (%7.3f ms/step, sweep %.4f, back-substitution %.4f, sweep %s) | streamed %8.0f = %.3f of resident (%.2f ms per batch, steady %.2f, submit %.2f, wait %.2f)" % (
        sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_backsub"]["avg_launch_ms"], d["roofline"].get("lba_elimination"),
        s.get("value", 0), s.get("fraction_of_resident", 0), s.get("ms_per_batch", 0), s.get("steady_ms_per_batch", 0), s.get("ms_per_batch_in_submit", 0), s.get("ms_per_batch_waiting_in_collect", 0)))
except Exception as e:
    print("windows %s FAILED %r" % (sys.argv[1], e))
PY
done
cat gpurun_out/${TAG}.txt
