#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/quick_tests.log
timeout 300 python tools/solve_phases.py > gpurun_out/quick_phases.log 2>&1
timeout 300 python tools/latency_sweep.py > gpurun_out/quick_latency.log 2>&1
SLSLAM_DEBUG_ABLATE=1024 timeout 300 python tools/latency_sweep.py > gpurun_out/quick_latency_nofuse.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
cat gpurun_out/quick_tests.log; tail -3 gpurun_out/quick_phases.log; head -4 gpurun_out/quick_latency.log; head -4 gpurun_out/quick_latency_nofuse.log; cut -c1-600 gpurun_out/quick_bench.json
