#!/bin/bash
# parity study (tolerances of the GPU tests) + the phase timing of the single-window reduced solve
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/parity_study.py > gpurun_out/parity_study.txt 2> gpurun_out/parity_study.err
timeout 300 python tools/solve_phases.py > gpurun_out/r2g_phases.log 2>&1
timeout 300 python tools/latency_sweep.py > gpurun_out/r2g_latency.log 2>&1
grep "^#" gpurun_out/parity_study.txt; tail -5 gpurun_out/parity_study.err; tail -20 gpurun_out/r2g_phases.log; tail -12 gpurun_out/r2g_latency.log
