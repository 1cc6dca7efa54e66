#!/bin/bash
# pose-graph path: GPU tests, then factorisation time per iteration (structured / dense fp64 / dense fp32) at 260 poses and on a loop-heavy graph
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-po}
timeout 1200 python -m pytest tests/test_gpu_po.py tests/test_gpu_lba.py -x -q -m gpu -k "po or beyond or oversize" > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.log
tail -n 4 gpurun_out/${TAG}_tests.log
python - <<'PY' 2>&1 | tee gpurun_out/${TAG}_po.txt
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from slslam_amd import capi, synth
for (N, loops) in ((260, 8), (260, 64), (520, 16)):
    g = synth.make_pose_graph(7, num_poses=N, num_loops=loops)
    res = {}
    for name, kw in (("structured", {}), ("dense_fp64", dict(po_dense_factor=1)), ("dense_fp32", dict(po_factor_fp32=1))):
        capi.po_solve(g, **kw)
        x, s, tm = capi.po_solve_timed(g, **kw)
        res[name] = x
        n = tm["unknowns"]; f = tm["factor_ms"]
        extra = ""
        if name != "structured": extra = " %.2f TFLOP/s" % (n ** 3 / 3.0 / (f * 1e-3) / 1e12)
        else: extra = " junction unknowns %d" % tm["junction_unknowns"]
        print("N=%d loops=%d %-11s total %.3f ms device, factorisation %.4f ms per iteration, unknowns %d, steps %d+%d, cost %.9e%s" % (
            N, loops, name, tm["total_ms"], f, n, s["num_successful_steps"], s["num_unsuccessful_steps"], s["final_cost"], extra))
    print("   max |x_structured - x_dense64| = %.3e   max |x_f32 - x_f64| = %.3e" % (np.abs(res["structured"] - res["dense_fp64"]).max(), np.abs(res["dense_fp32"] - res["dense_fp64"]).max()))
PY
