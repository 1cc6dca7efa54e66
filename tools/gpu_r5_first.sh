#!/bin/bash
# round 5, first call: the headline-path oracle test, the pose-graph tests (po_api.hip changed), host<->device copy rates, the quick bench line
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_first}
(timeout 900 python -m pytest tests/test_gpu_lba.py -m gpu -x -q -s -k "headline" 2>&1 | tail -12) > gpurun_out/${TAG}_headline.log
(timeout 600 python -m pytest tests/test_gpu_po.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/${TAG}_po.log
timeout 300 python tools/pcie_probe.py > gpurun_out/${TAG}_pcie.json 2>&1
nproc > gpurun_out/${TAG}_host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/${TAG}_host.txt 2>&1; uptime >> gpurun_out/${TAG}_host.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_headline.log gpurun_out/${TAG}_po.log gpurun_out/${TAG}_pcie.json gpurun_out/${TAG}_host.txt; cut -c1-600 gpurun_out/${TAG}_bench.json
