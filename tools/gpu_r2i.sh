#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/latency_sweep.py > gpurun_out/r2i_latency.log 2>&1
SLSLAM_DEBUG_ABLATE=1024 timeout 300 python tools/latency_sweep.py > gpurun_out/r2i_latency_nofuse.log 2>&1
head -3 gpurun_out/r2i_latency.log; head -3 gpurun_out/r2i_latency_nofuse.log
