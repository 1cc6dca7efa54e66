#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_c}
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1
timeout 900 python tools/mixed_precision_study.py > gpurun_out/${TAG}_mixed_study.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -k "c_level" 2>&1 | tail -15) > gpurun_out/${TAG}_disttest.log
tail -14 gpurun_out/${TAG}_pack_bench.txt; tail -8 gpurun_out/${TAG}_mixed_study.txt; cat gpurun_out/${TAG}_disttest.log
