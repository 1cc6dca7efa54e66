#!/bin/bash
# quick GPU check of a kernel change: the GPU tests, then the default bench workload without the CPU legs
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-quick}
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/${TAG}_tests.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_tests.log; cut -c1-1500 gpurun_out/${TAG}_bench.json
