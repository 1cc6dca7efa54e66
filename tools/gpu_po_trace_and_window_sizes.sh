#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=po
for V in structured dense; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_po_$V -o t -- python tools/po_prof.py $V > gpurun_out/${TAG}_po_$V.log 2>&1
  python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_po_$V/*.db | head -1) > gpurun_out/${TAG}_po_kernel_trace_$V.txt 2>&1
  rm -rf gpurun_out/${TAG}_po_$V
done
head -30 gpurun_out/${TAG}_po_kernel_trace_structured.txt; head -20 gpurun_out/${TAG}_po_kernel_trace_dense.txt
python - <<'PY'
import time, numpy as np
from slslam_amd import capi, synth
for (kf, free, lines, mt) in ((80, 40, 1500, 30.0), (40, 20, 1500, 12.0)):
    w = synth.make_window(5, num_lines=lines, num_kf=kf, num_free=free, mean_track=mt)
    capi.lba_solve(w)
    t0 = time.perf_counter(); x, s, t = capi.lba_solve(w); dt = time.perf_counter() - t0
    print("W-size window kf=%d free=%d lines=%d obs=%d: %.2f ms per one-shot solve, steps %d+%d" % (kf, free, lines, len(w["camera_index"]), 1e3 * dt, s["num_successful_steps"], s["num_unsuccessful_steps"]))
PY
