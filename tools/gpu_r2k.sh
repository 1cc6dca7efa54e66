#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_po.py tests/test_gpu_lba.py tests/test_house_study.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r2k_tests.log
cat gpurun_out/r2k_tests.log
TAG=r2k
for V in structured dense; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_po_$V -o t -- python tools/po_prof.py $V > gpurun_out/${TAG}_po_$V.log 2>&1
  python tools/rocpd_summary.py $(ls gpurun_out/${TAG}_po_$V/*.db | head -1) > gpurun_out/${TAG}_po_kernel_trace_$V.txt 2>&1
  rm -rf gpurun_out/${TAG}_po_$V
done
head -8 gpurun_out/${TAG}_po_kernel_trace_structured.txt; head -6 gpurun_out/${TAG}_po_kernel_trace_dense.txt
timeout 300 python tools/po_bench.py 2>&1 | tail -5
