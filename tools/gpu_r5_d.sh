#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_d}
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1
timeout 900 python tools/mixed_precision_study.py > gpurun_out/${TAG}_mixed_study.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_lba.py -m gpu -q -x -s -k "mixed_precision" 2>&1 | tail -12) > gpurun_out/${TAG}_mixed.log
tail -9 gpurun_out/${TAG}_pack_bench.txt | head -6; tail -6 gpurun_out/${TAG}_mixed_study.txt; cat gpurun_out/${TAG}_mixed.log
